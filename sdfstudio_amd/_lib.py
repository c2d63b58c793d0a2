"""ctypes binding of libsdfhip.so (the C ABI declared in include/sdfhip.h).

There is NO fallback: if the shared object is missing or a call fails, an exception is raised.  The HIP
extension is the product; PyTorch only provides device memory, streams and torch.distributed.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDFHIP_LIB", os.path.join(_HERE, "libsdfhip.so"))  # override: A/B runs of two builds in one process tree

c_float_p = ctypes.c_void_p  # device pointers travel as raw addresses
c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32
c_f32 = ctypes.c_float


class GridCfg(ctypes.Structure):
    _fields_ = [
        ("n_levels", c_i32), ("n_features", c_i32), ("log2_hashmap_size", c_i32), ("base_resolution", c_i32),
        ("per_level_scale", c_f32), ("smoothstep", c_i32),
    ]


class GridLevel(ctypes.Structure):
    _fields_ = [
        ("scale", c_f32), ("resolution", ctypes.c_uint32), ("size", ctypes.c_uint32), ("offset", ctypes.c_uint32),
        ("hashed", ctypes.c_uint32),
    ]


class FieldCfg(ctypes.Structure):
    _fields_ = [
        ("num_layers", c_i32), ("hidden_dim", c_i32), ("geo_feat_dim", c_i32), ("num_layers_color", c_i32),
        ("hidden_dim_color", c_i32), ("skip_layer", c_i32), ("pe_degree", c_i32), ("use_position_encoding", c_i32),
        ("appearance_dim", c_i32), ("contract", c_i32), ("rgb_padding", c_f32), ("grid", GridCfg),
        ("activation", c_i32), ("skip_style", c_i32),  # background fields: ReLU network, MLP-style skip (zero = the SDF field)
        ("ref_flags", c_i32), ("pe_off_axis", c_i32),  # ref-nerf colour options (REF_*), NeRFEncoding(off_axis=True) for the position
    ]


REF_DIFFUSE, REF_TINT, REF_REFLECT, REF_NDOTV = 1, 2, 4, 8


MODE_SDF, MODE_GEO, MODE_FULL = 0, 1, 2

_SIGNATURES = {
    "sdfhip_version": (c_i32, []),
    "sdfhip_last_error": (ctypes.c_char_p, []),
    "sdfhip_padded_points": (c_i64, [c_i64]),
    "sdfhip_grid_levels": (c_i32, [ctypes.POINTER(GridCfg), ctypes.POINTER(GridLevel), ctypes.POINTER(c_i64)]),
    "sdfhip_field_create": (c_i32, [ctypes.POINTER(FieldCfg), ctypes.POINTER(ctypes.c_void_p)]),
    "sdfhip_field_destroy": (None, [ctypes.c_void_p]),
    "sdfhip_field_theta_size": (c_i64, [ctypes.c_void_p]),
    "sdfhip_field_num_linear": (c_i32, [ctypes.c_void_p]),
    "sdfhip_field_theta_layout": (c_i32, [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                          ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "sdfhip_field_table_size": (c_i64, [ctypes.c_void_p]),
    "sdfhip_field_backward_feat": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, ctypes.c_void_p, c_float_p, c_float_p,
                                           c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_refnerf_workspace_size": (c_i64, [c_i64, c_i32]),
    "sdfhip_refnerf_forward": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_f32, c_float_p,
                                       ctypes.c_void_p]),
    "sdfhip_refnerf_backward": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_f32, c_float_p,
                                        ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_field_packed_size": (c_i64, [ctypes.c_void_p]),
    "sdfhip_field_workspace_size": (c_i64, [ctypes.c_void_p, c_i64, c_i32]),
    "sdfhip_field_pack": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_field_weightnorm_rows": (c_i64, [ctypes.c_void_p]),
    "sdfhip_field_theta_from_weightnorm": (c_i32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_i32, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_field_theta_backward_weightnorm": (c_i32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_i32, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_i32,
                                                       ctypes.c_void_p]),
    "sdfhip_surface_loss_workspace_floats": (c_i64, []),
    "sdfhip_surface_loss_forward": (c_i32, [c_float_p, c_float_p, c_i64, c_float_p, c_float_p, c_float_p, c_f32, c_i64, c_float_p, c_float_p,
                                            ctypes.POINTER(c_f32), c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_surface_loss_backward": (c_i32, [c_float_p, c_float_p, c_i64, c_float_p, c_float_p, c_float_p, c_f32, c_i64, c_float_p, c_float_p,
                                             ctypes.POINTER(c_f32), ctypes.POINTER(ctypes.c_void_p), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_field_forward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                     c_i64, c_i32, c_float_p, c_i32, c_i32, ctypes.c_void_p, c_float_p, c_float_p,
                                     c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_field_backward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, ctypes.c_void_p,
                                      c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_geo_workspace_size": (c_i64, [ctypes.c_void_p, c_i64]),
    "sdfhip_geo_forward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, ctypes.c_void_p, c_float_p,
                                   c_float_p, ctypes.c_void_p]),
    "sdfhip_geo_backward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_i64, ctypes.c_void_p, c_float_p, c_float_p, c_float_p,
                                    c_float_p, ctypes.c_void_p]),
    "sdfhip_geo_forward_n": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i64, ctypes.c_void_p, c_float_p,
                                     c_float_p, ctypes.c_void_p]),
    "sdfhip_geo_backward_n": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_i64, ctypes.c_void_p, c_i64, c_float_p, c_float_p, c_float_p,
                                      c_float_p, ctypes.c_void_p]),
    "sdfhip_geo_forward_rays": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64,
                                        c_i32, ctypes.c_void_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_generate_rays": (c_i32, [c_float_p, c_float_p, c_float_p, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_f32, c_i64, c_float_p,
                                     c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sdfhip_color_workspace_size": (c_i64, [ctypes.c_void_p, c_i64]),
    "sdfhip_color_forward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32,
                                     ctypes.c_void_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_color_backward": (c_i32, [ctypes.c_void_p, c_float_p, c_i64, c_i32, ctypes.c_void_p, c_float_p, c_float_p, c_float_p,
                                      c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_sh4_embed": (c_i32, [c_float_p, c_float_p, c_i64, c_i32, c_float_p, ctypes.c_void_p]),
    "sdfhip_embedding_backward": (c_i32, [ctypes.c_void_p, c_float_p, c_i64, c_i32, c_i64, c_float_p, ctypes.c_void_p]),
    "sdfhip_numfield_workspace_size": (c_i64, [ctypes.c_void_p, c_i64]),
    "sdfhip_numfield_inference_workspace_size": (c_i64, [ctypes.c_void_p, c_i64]),
    "sdfhip_field_set_table_grad_callback": (c_i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sdfhip_numfield_sdf_rows": (c_i64, [c_i64]),
    "sdfhip_numfield_forward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32,
                                        c_float_p, ctypes.c_float, c_i32, ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                        ctypes.c_void_p]),
    "sdfhip_numfield_backward": (c_i32, [ctypes.c_void_p, c_float_p, c_float_p, c_i64, c_i32, ctypes.c_float, ctypes.c_void_p, c_float_p,
                                         c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_grid_encode_forward": (c_i32, [ctypes.POINTER(GridCfg), c_float_p, c_float_p, c_i64, c_float_p, ctypes.c_void_p]),
    "sdfhip_grid_encode_backward": (c_i32, [ctypes.POINTER(GridCfg), c_float_p, c_i64, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_grid_cell_dump": (c_i32, [ctypes.POINTER(GridCfg), c_float_p, c_i64, ctypes.c_void_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_proposal_forward": (c_i32, [ctypes.POINTER(GridCfg), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                        c_float_p, c_float_p, c_i64, c_i32, c_i32, c_float_p, ctypes.c_void_p]),
    "sdfhip_proposal_workspace_size": (c_i64, []),
    "sdfhip_proposal_backward": (c_i32, [ctypes.POINTER(GridCfg), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                         c_float_p, c_float_p, c_i64, c_i32, c_i32, c_float_p, ctypes.c_void_p, c_float_p,
                                         c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_sample_spaced": (c_i32, [c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_float_p, c_float_p, c_float_p,
                                     ctypes.c_void_p]),
    "sdfhip_sample_uniform": (c_i32, [c_float_p, c_float_p, c_float_p, c_i32, c_i64, c_i32, c_float_p, c_float_p, c_float_p,
                                      ctypes.c_void_p]),
    "sdfhip_sample_spacing": (c_i32, [c_i32, c_float_p, c_float_p, c_float_p, c_i32, c_i64, c_i32, c_float_p, c_float_p, c_float_p,
                                      ctypes.c_void_p]),
    "sdfhip_adam_step": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_f32,
                                 ctypes.c_void_p]),
    "sdfhip_adamw_step": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_f32,
                                 ctypes.c_void_p]),
    "sdfhip_sample_pdf_spacing": (c_i32, [c_i32, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i32, c_i64, c_i32, c_i32, c_f32,
                                          c_f32, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_surface_root": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_f32, ctypes.c_void_p, c_float_p,
                                    c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_volsdf_render_forward": (c_i32, [c_float_p] * 7 + [c_i64, c_i32] + [c_float_p] * 9 + [ctypes.c_void_p]),
    "sdfhip_volsdf_render_backward": (c_i32, [c_float_p] * 7 + [c_i64, c_i32] + [c_float_p] * 16 + [ctypes.c_void_p]),
    "sdfhip_march_count": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, c_i64, c_i32,
                                   c_f32, ctypes.c_void_p, ctypes.c_void_p]),
    "sdfhip_march_write": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, c_i64, c_i32,
                                   c_f32, ctypes.c_void_p, ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_march_count_dev": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, c_i64, c_i32,
                                       c_f32, c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sdfhip_march_write_capped": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, c_i64, c_i32,
                                          c_f32, c_float_p, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_packed_resample": (c_i32, [c_float_p, c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i32, ctypes.c_void_p, c_float_p,
                                       c_float_p, ctypes.c_void_p]),
    "sdfhip_packed_weights_forward": (c_i32, [c_float_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_packed_weights_backward": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, c_float_p,
                                               ctypes.c_void_p]),
    "sdfhip_packed_accumulate": (c_i32, [c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_void_p, c_i64, c_i32, c_float_p, ctypes.c_void_p]),
    "sdfhip_interlevel_terms": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_i32, c_f32, c_float_p, c_float_p,
                                        c_float_p, ctypes.c_void_p]),
    "sdfhip_sample_pdf_uniform": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i32, c_i64, c_i32, c_i32, c_f32,
                                          c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_merge_uniform": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_i32, c_float_p, ctypes.c_void_p,
                                     c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_volsdf_bound_step": (c_i32, [c_float_p, c_float_p, c_float_p, ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                         c_i64, c_i32, c_i32, c_f32, c_i32, c_float_p, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "sdfhip_neus_upsample": (c_i32, [c_float_p, c_float_p, c_float_p, ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_i32, c_i64, c_i32,
                                     c_i32, c_i32, c_f32, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p,
                                     c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_sample_pdf": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_i32, c_f32, c_f32,
                                  c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_density_weights_forward": (c_i32, [c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_float_p, ctypes.c_void_p]),
    "sdfhip_density_weights_backward": (c_i32, [c_float_p, c_float_p, c_float_p, c_i64, c_i32, c_float_p, c_float_p,
                                                ctypes.c_void_p]),
    "sdfhip_neus_render_forward": (c_i32, [c_float_p] * 8 + [c_f32, c_i64, c_i32] + [c_float_p] * 8 + [ctypes.c_void_p]),
    "sdfhip_neus_render_backward": (c_i32, [c_float_p] * 8 + [c_f32, c_i64, c_i32] + [c_float_p] * 14 + [ctypes.c_void_p]),
    "sdfhip_neus_render_bg_forward": (c_i32, [c_float_p] * 8 + [c_f32, c_i64, c_i32] + [c_float_p] * 12 + [ctypes.c_void_p]),
    "sdfhip_neus_render_bg_backward": (c_i32, [c_float_p] * 8 + [c_f32, c_i64, c_i32] + [c_float_p] * 19 + [ctypes.c_void_p]),
    "sdfhip_mono_depth_loss_forward": (c_i32, [c_float_p, c_float_p, c_i64, c_i32, c_f32, c_f32, c_f32, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_mono_depth_loss_backward": (c_i32, [c_float_p, c_float_p, c_i64, c_i32, c_f32, c_f32, c_f32, c_float_p, c_float_p, c_float_p,
                                                ctypes.c_void_p]),
    "sdfhip_fg_mask_loss_forward": (c_i32, [c_float_p, c_float_p, c_i64, c_f32, c_float_p, ctypes.c_void_p]),
    "sdfhip_fg_mask_loss_backward": (c_i32, [c_float_p, c_float_p, c_i64, c_f32, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_sensor_depth_loss_workspace_size": (ctypes.c_size_t, []),
    "sdfhip_sensor_depth_loss_forward": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i64, c_f32, ctypes.c_void_p,
                                                 c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_sensor_depth_loss_backward": (c_i32, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_i64, c_i64, c_f32, c_float_p,
                                                  c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "sdfhip_profile_enable": (c_i32, [c_i32]),
    "sdfhip_profile_enable_slots": (c_i32, [ctypes.c_uint64]),
    "sdfhip_profile_name": (ctypes.c_char_p, [c_i32]),
    "sdfhip_profile_read": (c_i32, [c_i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
}

_lib: Optional[ctypes.CDLL] = None


class SdfHipError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES.keys())


def load() -> ctypes.CDLL:
    """Load libsdfhip.so; raises if it has not been built (python -m sdfstudio_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdfHipError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m sdfstudio_amd.build or "
            "__graft_entry__.build()). There is no CPU / PyTorch fallback for this path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sdfhip_last_error()
        raise SdfHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]):
    """Device address of a contiguous fp32 CUDA(HIP) tensor, or NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SdfHipError("sdfhip kernels need HIP device tensors (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise SdfHipError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    if t.numel() == 0:
        return _empty_ptr(t.device)
    return ctypes.c_void_p(t.data_ptr())


def rawptr(t: torch.Tensor):
    """Device address of a workspace / index tensor of any dtype (no layout checks); the valid dummy address for an empty one."""
    if t.numel() == 0:
        return _empty_ptr(t.device)
    return ctypes.c_void_p(t.data_ptr())


_EMPTY = {}


def _empty_ptr(device):
    """An EMPTY tensor has data_ptr() == 0, which the entry points would take for a missing (NULL) argument.  A present-but-empty
    argument (zero rays in an eval chunk, zero packed samples) is passed as a valid address nobody dereferences: the entry points return
    before launching when their element count is zero, as torch operators do on empty tensors."""
    key = (device.type, device.index)
    buf = _EMPTY.get(key)
    if buf is None:
        buf = _EMPTY[key] = torch.zeros(64, dtype=torch.float32, device=device)
    return ctypes.c_void_p(buf.data_ptr())


class Keep:
    """Marshals optional tensors into device pointers for ONE native call and keeps what it hands out alive.

    ``kp = Keep(); lib.fn(kp(a), kp(b), ...); del kp`` - a ``.contiguous()`` copy made for argument i must not be freed
    before argument i + 1 is evaluated (the caching allocator would hand the same block to the next copy and the two
    pointers alias); kernels are stream ordered, so keeping the copies until the launch has been issued is enough.
    """

    def __init__(self):
        self._refs = []

    def __call__(self, t: Optional[torch.Tensor]):
        if t is None:
            return None
        t = t.contiguous()
        self._refs.append(t)
        return ptr(t)


def ptr_array(tensors):
    """HOST array of device pointers (NULL for None) for the entry points that take `const float* const*`; the tensors must stay
    alive (and contiguous fp32 on the device) until the launch has been issued - keep them in a local."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else ptr(t)
    return arr


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def padded_points(n: int) -> int:
    return (n + 127) // 128 * 128


def grid_levels(cfg: GridCfg):
    """Host-only query (works without a GPU)."""
    lib = load()
    levels = (GridLevel * cfg.n_levels)()
    n = c_i64(0)
    check(lib.sdfhip_grid_levels(ctypes.byref(cfg), levels, ctypes.byref(n)), "sdfhip_grid_levels")
    return list(levels), int(n.value)


TABLE_GRAD_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
_table_grad_cb_keepalive = {}  # field handle value -> the ctypes thunk the C side holds a raw pointer to


def field_set_table_grad_callback(handle, fn) -> None:
    """fn(table_bar_ptr: int, stream_ptr: int) is called from inside sdfhip_field_backward / sdfhip_numfield_backward OF THIS FIELD (handle:
    SDFField._handle) right after the hash-table scatter has been enqueued on `stream` (include/sdfhip.h:
    sdfhip_field_set_table_grad_callback); None clears it.  Exceptions cannot cross the C frame: the callee has to catch and report them."""
    lib = load()
    key = int(handle.value if hasattr(handle, "value") else handle)
    if fn is None:
        check(lib.sdfhip_field_set_table_grad_callback(handle, None, None), "field_set_table_grad_callback")
        _table_grad_cb_keepalive.pop(key, None)
        return
    cb = TABLE_GRAD_CB(lambda _user, table_bar, stream: fn(int(table_bar or 0), int(stream or 0)))
    check(lib.sdfhip_field_set_table_grad_callback(handle, ctypes.cast(cb, ctypes.c_void_p), None), "field_set_table_grad_callback")
    _table_grad_cb_keepalive[key] = cb


def profile_enable(on: bool) -> int:
    return load().sdfhip_profile_enable(1 if on else 0)


def profile_enable_only(names) -> int:
    """HIP events on the launches of the named slots only (sdfhip_profile_name); every other launch runs un-instrumented."""
    lib = load()
    mask = 0
    n_slots = lib.sdfhip_profile_enable(0)
    for slot in range(n_slots):
        if lib.sdfhip_profile_name(slot).decode() in names:
            mask |= 1 << slot
    if not mask:
        raise ValueError(f"no profile slot named any of {list(names)}")
    return lib.sdfhip_profile_enable_slots(mask)


def profile_collect():
    lib = load()
    out = {}
    n_slots = 17
    for slot in range(n_slots):
        ms = ctypes.c_double(0.0)
        cnt = c_i64(0)
        check(lib.sdfhip_profile_read(slot, ctypes.byref(ms), ctypes.byref(cnt)), "profile_read")
        if cnt.value > 0:
            out[lib.sdfhip_profile_name(slot).decode()] = (ms.value, int(cnt.value))
    return out
