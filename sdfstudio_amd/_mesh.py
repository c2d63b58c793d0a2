"""ctypes binding of libsdfmesh.so (the C ABI declared in include/sdfmesh.h): marching cubes on a device-resident volume.

No fallback, as _lib.py: a missing library or a failing call raises.  PyTorch provides device memory and the stream.
"""
import ctypes
import os
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDFMESH_LIB", os.path.join(_HERE, "libsdfmesh.so"))

_p = ctypes.c_void_p
_SIGNATURES = {
    "sdfmesh_version": (ctypes.c_int, []),
    "sdfmesh_last_error": (ctypes.c_char_p, []),
    "sdfmesh_mc_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sdfmesh_mc_count": (ctypes.c_int, [_p, _p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _p]),
    "sdfmesh_mc_emit": (ctypes.c_int, [_p, _p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _p, ctypes.c_size_t,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _p, _p, _p, _p, _p]),
}
_lib: Optional[ctypes.CDLL] = None


class SdfMeshError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libsdfmesh.so; raises if it has not been built (python -m sdfstudio_amd.build / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdfMeshError(f"{LIB_PATH} is missing: build the HIP extensions first (python -m sdfstudio_amd.build). "
                           "There is no CPU fallback for the mesh path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return (load().sdfmesh_last_error() or b"").decode(errors="replace")


def _check(rc: int, what: str):
    if rc != 0:
        raise SdfMeshError(f"{what} failed ({rc}): {last_error()}")


def marching_cubes_device(volume: torch.Tensor, level: float, mask: Optional[torch.Tensor] = None, flip_faces: bool = True,
                          with_normals: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """sdfmesh_mc_count + sdfmesh_mc_emit on a CUDA float32 volume [n0, n1, n2]: (verts [V,3] float32 in lattice units and volume
    axis order, faces [F,3] int32, normals [V,3] float32 | None, values [V] float32 | None), all on the volume's device, in
    scikit-image's array order.  The one host synchronisation is the mesh size (include/sdfmesh.h)."""
    lib = load()
    if not volume.is_cuda:
        raise SdfMeshError("marching_cubes_device: the volume must live on the GPU (there is no CPU path)")
    if volume.dim() != 3:
        raise ValueError("Input volume should be a 3D array.")
    vol = volume.contiguous().float()
    n0, n1, n2 = (int(s) for s in vol.shape)
    m = None
    if mask is not None:
        if tuple(mask.shape) != tuple(vol.shape):
            raise ValueError("volume and mask must have the same shape.")
        m = (mask.to(device=vol.device) != 0).contiguous().to(torch.uint8)  # truthiness, as scikit-image reads its boolean mask
    need = int(lib.sdfmesh_mc_workspace_bytes(n0, n1, n2))
    if need == 0:
        raise SdfMeshError(f"marching_cubes_device: volume {tuple(vol.shape)} refused: {last_error() or 'every dimension >= 2, < 2^31 points'}")
    with torch.cuda.device(vol.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws = torch.empty(need, dtype=torch.uint8, device=vol.device)
        nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib.sdfmesh_mc_count(vol.data_ptr(), None if m is None else m.data_ptr(), n0, n1, n2, float(level), ws.data_ptr(), need,
                                    ctypes.byref(nv), ctypes.byref(nf), stream), "sdfmesh_mc_count")
        V, F = int(nv.value), int(nf.value)
        verts = torch.empty(V, 3, dtype=torch.float32, device=vol.device)
        faces = torch.empty(F, 3, dtype=torch.int32, device=vol.device)
        normals = torch.empty(V, 3, dtype=torch.float32, device=vol.device) if with_normals else None
        values = torch.empty(V, dtype=torch.float32, device=vol.device) if with_normals else None
        if V or F:
            _check(lib.sdfmesh_mc_emit(vol.data_ptr(), None if m is None else m.data_ptr(), n0, n1, n2, float(level), ws.data_ptr(), need,
                                       V, F, 1 if flip_faces else 0, verts.data_ptr(), faces.data_ptr(),
                                       None if normals is None else normals.data_ptr(), None if values is None else values.data_ptr(), stream),
                   "sdfmesh_mc_emit")
    return verts, faces, normals, values
