"""The "grid" background field of the surface models: nerfstudio/fields/nerfacto_field.py:65-332 (TCNNNerfactoField) as
base_surface_model.py:181-187 builds it for BASELINE config 5 (neus-facto-angelo / bakedangelo, method_configs.py:423): a
16-level hash grid + 2-layer 64-wide ReLU MLP -> (density, 15 geometry features), then spherical harmonics of the view direction
+ features + appearance embedding -> 3-layer 64-wide ReLU MLP -> sigmoid rgb.

The reference builds every piece from tiny-cuda-nn (HashGrid, FullyFusedMLP without biases, SphericalHarmonics degree 4); here the
whole field runs on the fused sdfhip kernels instantiated with a ReLU activation (csrc/inst_e.hip): hash-grid gather +
base MLP + density row are one ``geo_fwd_kernel`` launch (``sdfhip_geo_forward``; backward: ``geo_bwd_kernel<TANGENT = false>``,
``grid_bwd_kernel``, split-K weight gradients), the colour MLP one ``col_fwd_kernel`` launch (``sdfhip_color_forward``) that takes the
16 spherical harmonics of the view direction and the appearance embedding - both per-ray quantities - in the per-ray embedding slots
of its small-input block.  The harmonics themselves are 16 polynomials per RAY (torch, [N, 16]).  No torch matmul, no rocBLAS, no CPU
path.  Optional heads of the reference (transients, semantics, predicted normals: all off in the surface models) raise.
``hash_grid_encode`` below is the encoding as a standalone operator (tcnn.Encoding("HashGrid")), kept for callers that want features.
tcnn keeps each network's weights in one fp16 ``params`` vector in an internal padded layout, so reference checkpoints of THIS field do
not load (fp32 ``mlp_base.{table, w1, w2}``, ``mlp_head.{w1, w2, w3}`` here), as for the proposal networks.
"""
import math
from typing import Dict

import torch
from torch import nn

from sdfstudio_amd import _lib
from sdfstudio_amd.fields.field_heads import FieldHeadNames


class _GridEncode(torch.autograd.Function):
    """tcnn.Encoding("HashGrid"): x [P,3] in [0,1] -> features [P, L*F]; gradient w.r.t. the table only."""

    @staticmethod
    def forward(ctx, table, x, cfg):
        lib = _lib.load()
        x = x.contiguous()
        feat = torch.empty(x.shape[0], cfg.n_levels * cfg.n_features, device=x.device)
        _lib.check(lib.sdfhip_grid_encode_forward(cfg, _lib.ptr(table), _lib.ptr(x), x.shape[0], _lib.ptr(feat), _lib.stream()),
                   "grid_encode_forward")
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n_table = cfg, table.numel()
        return feat

    @staticmethod
    def backward(ctx, fbar):
        (x,) = ctx.saved_tensors
        lib = _lib.load()
        tbar = torch.zeros(ctx.n_table, device=x.device)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_grid_encode_backward(ctx.cfg, _lib.ptr(x), x.shape[0], kp(fbar), _lib.ptr(tbar), _lib.stream()),
                   "grid_encode_backward")
        del kp
        return tbar, None, None


def hash_grid_encode(table: torch.Tensor, x: torch.Tensor, cfg) -> torch.Tensor:
    return _GridEncode.apply(table, x, cfg)


def sh_degree4(d: torch.Tensor) -> torch.Tensor:
    """tiny-cuda-nn's SphericalHarmonics encoding, degree 4 (16 real harmonics), on directions given in [0,1]^3 (the encoding maps
    them back to [-1,1]^3 itself; nerfacto_field.py:128-134 feeds get_normalized_directions(d) = (d + 1) / 2)."""
    v = d * 2.0 - 1.0
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ], dim=-1)


class _ShEmbed(torch.autograd.Function):
    """[SH degree 4 of (d + 1) / 2 | appearance embedding] per ray: the per-ray inputs of the field's colour network in one native
    launch (sdfhip_sh4_embed); sh_degree4 above is the same statement in torch ops (~40 launches)."""

    @staticmethod
    def forward(ctx, dirs, emb, emb_dim):
        lib = _lib.load()
        n = dirs.shape[0]
        out = torch.empty(n, 16 + emb_dim, device=dirs.device)
        emb_c = None if emb is None else emb.contiguous()
        _lib.check(lib.sdfhip_sh4_embed(_lib.ptr(dirs), _lib.ptr(emb_c), n, emb_dim, _lib.ptr(out), _lib.stream()), "sh4_embed")
        del emb_c
        return out

    @staticmethod
    def backward(ctx, out_bar):
        return None, (out_bar[:, 16:] if ctx.needs_input_grad[1] else None), None


class _PermutedTheta(torch.autograd.Function):
    """theta = cat(parameters..., one zero)[src] where src visits every parameter element exactly once (a permutation with zero padding):
    forward = one cat + one gather; backward = one gather of theta_bar per parameter through the INVERSE map, written straight into the
    parameter's slot of the flat gradient buffer (sdfstudio_amd/grad_slots.py).  Autograd's own backward of the same two statements is an
    index_add_ over the whole vector (0.06 ms), a split, and one copy per parameter into its slot."""

    @staticmethod
    def forward(ctx, src, zero, invs, *params):
        ctx.params, ctx.invs = params, invs
        return torch.cat([p.reshape(-1) for p in params] + [zero]).index_select(0, src)

    @staticmethod
    def backward(ctx, theta_bar):
        from sdfstudio_amd.grad_slots import grad_target

        theta_bar = theta_bar.contiguous()
        outs = []
        for p, inv in zip(ctx.params, ctx.invs):
            out = grad_target(p)[0]  # every element is written
            torch.index_select(theta_bar, 0, inv, out=out.view(-1))
            outs.append(out)
        return (None, None, None, *outs)


class _TruncExp(torch.autograd.Function):
    """field_components/activations.py:23-39: exp forward, gradient exp(clamp(x, -15, 15))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def _xavier(o: int, i: int) -> nn.Parameter:
    a = math.sqrt(6.0 / (i + o))
    return nn.Parameter((torch.rand(o, i) * 2 - 1) * a)  # tcnn FullyFusedMLP: xavier_uniform, no biases


class _BaseParams(nn.Module):
    def __init__(self, n_table: int, d_in: int, hidden: int, d_out: int):
        super().__init__()
        self.table = nn.Parameter((torch.rand(n_table) * 2 - 1) * 1e-4)  # tcnn grid init U(-1e-4, 1e-4)
        self.w1, self.w2 = _xavier(hidden, d_in), _xavier(d_out, hidden)


class _HeadParams(nn.Module):
    def __init__(self, d_in: int, hidden: int, d_out: int):
        super().__init__()
        self.w1, self.w2, self.w3 = _xavier(hidden, d_in), _xavier(hidden, hidden), _xavier(d_out, hidden)


class TCNNNerfactoField(nn.Module):
    """nerfacto_field.py:65-332 with the defaults base_surface_model.py:181-187 relies on."""

    def __init__(self, aabb, num_images: int, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15, num_levels: int = 16,
                 max_res: int = 1024, log2_hashmap_size: int = 19, num_layers_color: int = 3, num_layers_transient: int = 2,
                 hidden_dim_color: int = 64, hidden_dim_transient: int = 64, appearance_embedding_dim: int = 32, transient_embedding_dim: int = 16,
                 use_transient_embedding: bool = False, use_semantics: bool = False, num_semantic_classes: int = 100,
                 use_pred_normals: bool = False, use_average_appearance_embedding: bool = False, spatial_distortion=None) -> None:
        super().__init__()
        if use_transient_embedding or use_semantics or use_pred_normals:
            raise NotImplementedError("transient / semantic / predicted-normal heads of TCNNNerfactoField are not built (off in the surface models)")
        if num_layers != 2 or num_layers_color != 3:
            raise NotImplementedError("built for the reference's defaults: 2-layer base MLP, 3-layer colour MLP")
        self.aabb = nn.Parameter(torch.as_tensor(aabb, dtype=torch.float32).clone(), requires_grad=False)  # a Parameter in the reference too (:110)
        self.geo_feat_dim, self.num_images, self.appearance_embedding_dim = geo_feat_dim, num_images, appearance_embedding_dim
        self.spatial_distortion = spatial_distortion
        self.use_average_appearance_embedding = use_average_appearance_embedding
        from sdfstudio_amd.fields.sdf_field import _Embedding

        self.embedding_appearance = _Embedding(num_images, appearance_embedding_dim)  # field_components/embedding.py (:116)
        base_res, features_per_level = 16, 2
        growth = math.exp((math.log(max_res) - math.log(base_res)) / (num_levels - 1))
        self.grid_cfg = _lib.GridCfg(num_levels, features_per_level, log2_hashmap_size, base_res, growth, 0)
        _, n_entries = _lib.grid_levels(self.grid_cfg)
        self.mlp_base = _BaseParams(n_entries * features_per_level, num_levels * features_per_level, hidden_dim, 1 + geo_feat_dim)
        self.mlp_head = _HeadParams(16 + geo_feat_dim + appearance_embedding_dim, hidden_dim_color, 3)
        # ---- native descriptor: a ReLU geometry-type network on in0 = [x (zero weights) | grid features], 1 hidden layer, output =
        # [density ; geometry features padded to one 32-wide block]; colour-type network with 2 hidden layers whose per-ray embedding
        # slots carry [SH(16) | appearance embedding]
        from sdfstudio_amd.fields.vanilla_nerf_field import NativeBackgroundNet

        if geo_feat_dim > 32 or hidden_dim % 32 or hidden_dim_color % 32:
            raise NotImplementedError("geo_feat_dim <= 32 and hidden widths that are multiples of 32")
        self._gf_pad = 32
        n_in, e_dim = num_levels * features_per_level, 16 + appearance_embedding_dim
        # scene contraction of the frustum mid points: applied by the ray entry of the geometry network (sdfhip_geo_forward_rays)
        order = getattr(spatial_distortion, "order", None) if spatial_distortion is not None else None
        self._contract = 0 if spatial_distortion is None else (1 if order == float("inf") else 2 if order in (None, 2) else -1)
        cfg_c = _lib.FieldCfg(1, hidden_dim, self._gf_pad, 2, hidden_dim_color, -1, 0, 0, e_dim, max(self._contract, 0), 0.0, self.grid_cfg, 1, 0)
        expect = [(hidden_dim, 3 + n_in), (1 + self._gf_pad, hidden_dim), (hidden_dim_color, 33 + self._gf_pad + e_dim),
                  (hidden_dim_color, hidden_dim_color), (3, hidden_dim_color)]
        self._native = NativeBackgroundNet(cfg_c, self._gf_pad, e_dim, expect)

    def _const(self, name: str, shape, value: float, device) -> torch.Tensor:
        """Constant device tensors the kernels read every step (level mask of ones, the unused normal input): built once per shape."""
        cache = self.__dict__.setdefault("_const_cache", {})
        key = (name, tuple(shape), str(device))
        if key not in cache:
            cache[key] = torch.full(tuple(shape), float(value), device=device)
        return cache[key]

    def _theta_layout(self) -> torch.Tensor:
        """The library's flat parameter vector as cat() statements over the field's tensors (tcnn's FullyFusedMLP has no biases: they are
        zeros here and their gradients are dropped).  Evaluated ONCE, on tagged stand-ins, to learn where every parameter element lands."""
        b, h = self.mlp_base, self.mlp_head
        H, HC, G, GP, E = b.w1.shape[0], h.w1.shape[0], self.geo_feat_dim, self._gf_pad, self.appearance_embedding_dim
        tags, off = {}, 0
        for name, t in (("bw1", b.w1), ("bw2", b.w2), ("hw1", h.w1), ("hw2", h.w2), ("hw3", h.w3)):
            tags[name] = torch.arange(off, off + t.numel(), dtype=torch.int64).view_as(t)
            off += t.numel()
        z = lambda *shape: torch.full(shape, off, dtype=torch.int64)  # noqa: E731  ("off" = the slot of an appended zero)
        bw1, bw2, hw1, hw2, hw3 = (tags[k] for k in ("bw1", "bw2", "hw1", "hw2", "hw3"))
        parts = [torch.cat([z(H, 3), bw1], dim=1).reshape(-1), z(H),
                 torch.cat([bw2, z(1 + GP - bw2.shape[0], H)], dim=0).reshape(-1), z(1 + GP),
                 # colour layer 0 columns: x(3) d-PE(27) normal(3) [all unused] | features (GP) | embedding slots = SH(16) + appearance (E)
                 torch.cat([z(HC, 33), hw1[:, 16:16 + G], z(HC, GP - G), hw1[:, :16], hw1[:, 16 + G:16 + G + E]], dim=1).reshape(-1), z(HC),
                 hw2.reshape(-1), z(HC), hw3.reshape(-1), z(3)]
        return torch.cat(parts)

    def _theta(self) -> torch.Tensor:
        """Flat parameter vector in the library's layout: ONE concatenation of the five weight tensors (plus a zero) and ONE gather
        through the index map of _theta_layout - the cat-of-cats statement itself cost ~60 small launches per step with its backward."""
        b, h = self.mlp_base, self.mlp_head
        dev = b.w1.device
        params = (b.w1, b.w2, h.w1, h.w2, h.w3)
        if getattr(self, "_theta_src", None) is None or self._theta_src.device != dev:
            src = self._theta_layout()
            n_src = sum(p.numel() for p in params)
            # inverse map: where in theta does source element j land?  (None when some element is used twice or never: autograd's own path)
            hits = torch.bincount(src[src < n_src], minlength=n_src)
            self._theta_invs = None
            if bool((hits == 1).all()):
                where = torch.empty(n_src, dtype=torch.int64)
                where[src[src < n_src]] = torch.nonzero(src < n_src)[:, 0]
                self._theta_invs, off = [], 0
                for p in params:
                    self._theta_invs.append(where[off:off + p.numel()].to(dev))
                    off += p.numel()
            self._theta_src = src.to(dev)
            self._theta_zero = torch.zeros(1, device=dev, dtype=b.w1.dtype)
        if self._theta_invs is not None and torch.is_grad_enabled():
            return _PermutedTheta.apply(self._theta_src, self._theta_zero, self._theta_invs, *params)
        flat = torch.cat([p.reshape(-1) for p in params] + [self._theta_zero])
        return flat.index_select(0, self._theta_src)

    def get_density(self, ray_samples):
        """:225-246: contracted frustum MID points -> (x + 2) / 4 -> hash grid -> MLP -> trunc_exp of the first output.  The second
        return value carries what get_outputs needs from this call (features + the packed parameter vector + positions)."""
        from sdfstudio_amd.cameras.rays import unpack_ray_samples
        from sdfstudio_amd.fields.sdf_field import _GeoNetFunction, _GeoNetRaysFunction

        if self._contract > 0 and getattr(ray_samples.frustums, "offsets", None) is None and ray_samples.frustums.starts.is_cuda:
            # positions from the frustums and their contraction inside the kernel
            o, d, st, en = unpack_ray_samples(ray_samples)
            theta = self._theta()
            mask = self._const("mask", (self.grid_cfg.n_levels * self.grid_cfg.n_features,), 1.0, st.device)
            pre, feat, x = _GeoNetRaysFunction.apply(theta, self.mlp_base.table, self._native, o.detach().float(), d.detach().float(),
                                                     st.detach().float(), en.detach().float(), mask)
            return _TruncExp.apply(pre.view(*st.shape, 1)), (feat, theta, x)
        positions = ray_samples.frustums.get_positions()
        shape = tuple(positions.shape[:-1])
        if self.spatial_distortion is not None:
            positions = self.spatial_distortion(positions)  # the kernel applies (x + 2) / 4 itself (sdf_field.py:384, same map)
        else:
            positions = (positions - self.aabb[0]) / (self.aabb[1] - self.aabb[0]) * 4.0 - 2.0  # SceneBox.get_normalized_positions
        x = positions.reshape(-1, 3).detach().float().contiguous()
        if not x.is_cuda:
            raise _lib.SdfHipError("TCNNNerfactoField runs on the sdfhip kernels: HIP device tensors required (no CPU fallback)")
        theta = self._theta()
        mask = self._const("mask", (self.grid_cfg.n_levels * self.grid_cfg.n_features,), 1.0, x.device)
        pre, feat = _GeoNetFunction.apply(theta, self.mlp_base.table, self._native, x, mask)
        density = _TruncExp.apply(pre.view(*shape, 1))
        return density, (feat, theta, x)

    def get_outputs(self, ray_samples, density_embedding=None) -> Dict:
        """:248-330."""
        from sdfstudio_amd.cameras.rays import unpack_ray_samples
        from sdfstudio_amd.fields.sdf_field import _ColorFunction

        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        feat, theta, x = density_embedding
        _, d, st, _ = unpack_ray_samples(ray_samples)
        n, s = st.shape
        if self.training:
            emb = self.embedding_appearance(ray_samples.camera_indices.reshape(n, -1)[:, 0])
        elif self.use_average_appearance_embedding:
            emb = self.embedding_appearance.mean(dim=0)[None, :].expand(n, -1)
        else:
            emb = None  # zeros
        slots = _ShEmbed.apply(d.contiguous(), emb, self.appearance_embedding_dim)  # [SH(get_normalized_directions(d)) | emb]: one launch
        # the colour kernel's normal input is unused by this field (zero weight columns): a cached zero block, not a fill per call
        rgb = _ColorFunction.apply(theta, feat, self._const("zero_normal", tuple(x.shape), 0.0, x.device), slots, self._native, x, d.contiguous(), n, s)
        return {FieldHeadNames.RGB: rgb.view(n, s, 3)}

    def density_fn(self, positions: torch.Tensor) -> torch.Tensor:
        """Field.density_fn (fields/base_field.py:48-65): the density at explicit positions [..., 3] (zero-length frustums at the positions)."""
        from sdfstudio_amd.cameras.rays import Frustums, RaySamples

        flat = positions.reshape(-1, 1, 3)
        one = torch.ones_like(flat[..., :1])
        rs = RaySamples(frustums=Frustums(origins=flat, directions=torch.ones_like(flat), starts=torch.zeros_like(one), ends=torch.zeros_like(one),
                                          pixel_area=one))
        density, _ = self.get_density(rs)
        return density.view(*positions.shape[:-1], 1)

    def forward(self, ray_samples, compute_normals: bool = False) -> Dict:
        """fields/base_field.py:111-126."""
        if compute_normals:
            raise NotImplementedError("compute_normals (normals of a density field, base_field.py:104-121) is not built")
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=emb)
        out[FieldHeadNames.DENSITY] = density
        return out
