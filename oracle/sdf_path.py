"""Oracle (test infrastructure): the NeuS-facto / SDF-field hot path restated in plain PyTorch.

Functional, CPU, dtype follows the inputs (fp32 for golden vectors, fp64 for
tie-breaking).  Each function cites the reference file:line it follows
(paths relative to ``/root/reference/nerfstudio``).  Parameters are passed as a flat
dict whose keys are the reference ``state_dict`` names (``glin0.weight_v`` ...), so a
reference module's ``state_dict()`` can be dropped in unchanged.

Tensor conventions differ from the reference on purpose (this is a restatement, not a
copy): per-sample scalars are ``[N,S]`` (the reference carries a trailing 1), ray
scalars are ``[N]``.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from oracle import hashgrid

Params = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- config
@dataclass
class FieldCfg:
    """Mirror of SDFFieldConfig (fields/sdf_field.py:121-185), only the knobs on the path."""

    num_layers: int = 8
    hidden_dim: int = 256
    geo_feat_dim: int = 256
    num_layers_color: int = 4
    hidden_dim_color: int = 256
    appearance_embedding_dim: int = 32
    use_appearance_embedding: bool = False
    bias: float = 0.5
    inside_outside: bool = False
    use_grid_feature: bool = True
    beta_init: float = 0.3
    position_encoding_max_degree: int = 6
    use_position_encoding: bool = True
    rgb_padding: float = 0.001
    use_numerical_gradients: bool = False
    num_levels: int = 16
    max_res: int = 2048
    base_res: int = 16
    log2_hashmap_size: int = 19
    hash_features_per_level: int = 2
    hash_smoothstep: bool = True
    skip_in: Tuple[int, ...] = (4,)
    # the ref-nerf options of get_colors and NeRFEncoding(off_axis) (sdf_field.py:154-162, 268; the bakedsdf / bakedangelo field settings,
    # configs/method_configs.py:270-286)
    use_diffuse_color: bool = False
    use_specular_tint: bool = False
    use_reflections: bool = False
    use_n_dot_v: bool = False
    off_axis: bool = False

    def growth_factor(self) -> float:
        # sdf_field.py:226
        return math.exp((math.log(self.max_res) - math.log(self.base_res)) / (self.num_levels - 1))

    def grid_levels(self) -> hashgrid.GridLevels:
        return hashgrid.make_levels(
            self.num_levels, self.hash_features_per_level, self.log2_hashmap_size, self.base_res,
            self.growth_factor(), self.hash_smoothstep,
        )

    def geo_in_dim(self) -> int:
        # encodings.py:165-175 get_out_dim: in_dim (or the 21 directions of P, off_axis) * num_frequencies * 2
        return 3 + (21 if self.off_axis else 3) * 2 * self.position_encoding_max_degree + self.num_levels * self.hash_features_per_level

    def geo_dims(self) -> List[int]:
        # sdf_field.py:279-282
        return [self.geo_in_dim()] + [self.hidden_dim] * self.num_layers + [1 + self.geo_feat_dim]

    def color_in_dim(self) -> int:
        # sdf_field.py:338-352 (point, view dir PE, normal, feature, embedding; use_diffuse_color: dir PE, feature, embedding; + n . v)
        base = 27 + self.geo_feat_dim + self.appearance_embedding_dim + (0 if self.use_diffuse_color else 6)
        return base + (1 if self.use_n_dot_v else 0)

    def color_dims(self) -> List[int]:
        return [self.color_in_dim()] + [self.hidden_dim_color] * self.num_layers_color + [3]


@dataclass
class ProposalCfg:
    """Mirror of HashMLPDensityField's ctor args (fields/density_fields.py:52-65) as used by neus_facto.py:59-64."""

    hidden_dim: int = 16
    num_levels: int = 5
    max_res: int = 64
    base_res: int = 16
    log2_hashmap_size: int = 17
    features_per_level: int = 2

    def grid_levels(self) -> hashgrid.GridLevels:
        g = math.exp((math.log(self.max_res) - math.log(self.base_res)) / (self.num_levels - 1))
        return hashgrid.make_levels(
            self.num_levels, self.features_per_level, self.log2_hashmap_size, self.base_res, g, False
        )


@dataclass
class ModelCfg:
    """The NeuS-facto knobs on the path (models/neus_facto.py:43-97, base_surface_model.py:69-134)."""

    field: FieldCfg = field(default_factory=FieldCfg)
    proposals: Tuple[ProposalCfg, ...] = (ProposalCfg(max_res=64), ProposalCfg(max_res=256))
    num_proposal_samples: Tuple[int, ...] = (256, 96)
    num_neus_samples: int = 128
    eikonal_loss_mult: float = 0.1
    interlevel_loss_mult: float = 1.0
    near: float = 0.5
    far: float = 4.5
    histogram_padding: float = 0.01


# ----------------------------------------------------------------------------- small pieces
def nerf_encoding(x: torch.Tensor, num_frequencies: int, include_input: bool) -> torch.Tensor:
    """field_components/encodings.py:167-208 with min_freq_exp=0, max_freq_exp=num_frequencies-1.

    Layout: [sin(x_d * 2^f) for d, f] ++ [sin(x_d * 2^f + pi/2) for d, f] (++ x).
    """
    freqs = 2.0 ** torch.arange(num_frequencies, dtype=x.dtype, device=x.device)
    scaled = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    enc = torch.sin(torch.cat([scaled, scaled + math.pi / 2.0], dim=-1))
    if include_input:
        enc = torch.cat([enc, x], dim=-1)
    return enc


# NeRFEncoding.P (field_components/encodings.py:139-163): the 21 directions of the off-axis encoding, rows as the reference lists them
OFF_AXIS_P = torch.tensor([
    [0.8506508, 0, 0.5257311], [0.809017, 0.5, 0.309017], [0.5257311, 0.8506508, 0], [1, 0, 0], [0.809017, 0.5, -0.309017],
    [0.8506508, 0, -0.5257311], [0.309017, 0.809017, -0.5], [0, 0.5257311, -0.8506508], [0.5, 0.309017, -0.809017], [0, 1, 0],
    [-0.5257311, 0.8506508, 0], [-0.309017, 0.809017, -0.5], [0, 0.5257311, 0.8506508], [-0.309017, 0.809017, 0.5],
    [0.309017, 0.809017, 0.5], [0.5, 0.309017, 0.809017], [0.5, -0.309017, 0.809017], [0, 0, 1], [-0.5, 0.309017, 0.809017],
    [-0.809017, 0.5, 0.309017], [-0.809017, 0.5, -0.309017]], dtype=torch.float64)


def nerf_encoding_off_axis(x: torch.Tensor, num_frequencies: int) -> torch.Tensor:
    """field_components/encodings.py:190-198 with off_axis=True, include_input=False: scaled = (x @ P)[..., None] * freqs."""
    freqs = 2.0 ** torch.arange(num_frequencies, dtype=x.dtype, device=x.device)
    proj = torch.matmul(x, OFF_AXIS_P.T.to(x))
    scaled = (proj[..., None] * freqs).reshape(*x.shape[:-1], -1)
    return torch.sin(torch.cat([scaled, scaled + math.pi / 2.0], dim=-1))


def contract_inf(x: torch.Tensor) -> torch.Tensor:
    """field_components/spatial_distortions.py:66-73 with order=inf (base_surface_model.py:148-155)."""
    mag = x.abs().amax(dim=-1, keepdim=True)
    safe = torch.where(mag >= 1, mag, torch.ones_like(mag))
    return torch.where(mag >= 1, (2.0 - 1.0 / safe) * (x / safe), x)


def contract_l2(x: torch.Tensor) -> torch.Tensor:
    """field_components/spatial_distortions.py:66-73 with order=None (torch.linalg.norm's default: the 2-norm) -
    scene_contraction_norm = "l2" (base_surface_model.py:150-151)."""
    mag = torch.linalg.norm(x, dim=-1, keepdim=True)
    safe = torch.where(mag >= 1, mag, torch.ones_like(mag))
    return torch.where(mag >= 1, (2.0 - 1.0 / safe) * (x / safe), x)


def fold_weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """torch.nn.utils.weight_norm(dim=0) as applied at sdf_field.py:312-313: W = g * v / ||v||_row."""
    return g * v / v.norm(dim=1, keepdim=True)


def linear_wn(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    if f"{name}.weight_v" in p:
        w = fold_weight_norm(p[f"{name}.weight_v"], p[f"{name}.weight_g"])
    else:
        w = p[f"{name}.weight"]
    return x @ w.t() + p[f"{name}.bias"]


def softplus100(x: torch.Tensor) -> torch.Tensor:
    return F.softplus(x, beta=100)


# ----------------------------------------------------------------------------- SDF field
def geo_network(x: torch.Tensor, p: Params, cfg: FieldCfg, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fields/sdf_field.py:380-410 forward_geonetwork.  x: [P,3] -> [P, 1+geo_feat_dim]."""
    if cfg.use_grid_feature:
        lv = cfg.grid_levels()
        table = p["encoding.params"].view(lv.n_entries, lv.n_features)
        feat = hashgrid.grid_encode((x + 2.0) / 4.0, table, lv)
        if mask is not None:
            feat = feat * mask.to(feat)
    else:
        feat = torch.zeros(x.shape[0], cfg.num_levels * cfg.hash_features_per_level, dtype=x.dtype)
    if cfg.off_axis:
        pe = nerf_encoding_off_axis(x, cfg.position_encoding_max_degree)
    else:
        pe = nerf_encoding(x, cfg.position_encoding_max_degree, include_input=False)
    if not cfg.use_position_encoding:
        pe = torch.zeros_like(pe)
    inputs = torch.cat([x, pe, feat], dim=-1)
    h = inputs
    n_lin = cfg.num_layers + 1
    for l in range(n_lin):
        if l in cfg.skip_in:
            h = torch.cat([h, inputs], dim=1) / math.sqrt(2)
        h = linear_wn(p, f"glin{l}", h)
        if l < n_lin - 1:
            h = softplus100(h)
    return h


def sdf_and_gradient(x: torch.Tensor, p: Params, cfg: FieldCfg, mask=None, create_graph=True):
    """fields/sdf_field.py:631-654: geonetwork under enable_grad, analytic d sdf / d x."""
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        h = geo_network(x, p, cfg, mask)
        sdf = h[:, :1]
        grad = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=create_graph, retain_graph=True)[0]
    return sdf[:, 0], h[:, 1:], grad


# Test instrumentation for the one non-smooth spot of the field: d relu(z) / dz jumps at z = 0, so two correct fp32
# implementations whose pre-activations differ by round-off can legitimately disagree on the gradient contribution of a
# (point, unit) pair with |z| ~ 1e-7.  relu_hook(record={}) captures the colour network's pre-activations; relu_hook(flip=
# {layer: bool mask}) evaluates the other branch at the flagged pairs (tests/helpers.py: assert_grads_close_mod_relu_flips).
# Key -1 is the curvature loss's |c| (neus_facto_loss): same mechanism, same reason.  relu_hook(force={layer: bool mask}) imposes a
# whole branch pattern: PyTorch's multi-threaded CPU reductions are not run-to-run reproducible at the last ulp, so "the other branch
# than this run's" is not a well-defined request at a knife edge - "this pattern" is.
RELU_HOOK = None


class relu_hook:
    def __init__(self, record=None, flip=None, force=None):
        self.cfg = {"record": record, "flip": flip, "force": force}

    def __enter__(self):
        global RELU_HOOK
        self.prev, RELU_HOOK = RELU_HOOK, self.cfg
        return self

    def __exit__(self, *exc):
        global RELU_HOOK
        RELU_HOOK = self.prev
        return False


def numerical_gradient(x: torch.Tensor, p: Params, cfg: FieldCfg, delta: float, mask=None):
    """fields/sdf_field.py:431-453 (use_numerical_gradients): central differences of the sdf at x +- delta e_k, in the (contracted)
    space x lives in.  Returns (gradients [P,3], the six tap values [6,P])."""
    offs = torch.tensor([[delta, 0.0, 0.0], [-delta, 0.0, 0.0], [0.0, delta, 0.0], [0.0, -delta, 0.0], [0.0, 0.0, delta],
                         [0.0, 0.0, -delta]], dtype=x.dtype)
    pts = (x[None, :, :] + offs[:, None, :]).reshape(-1, 3)
    taps = geo_network(pts, p, cfg, mask)[:, 0].view(6, x.shape[0])
    grad = torch.stack([0.5 * (taps[0] - taps[1]) / delta, 0.5 * (taps[2] - taps[3]) / delta, 0.5 * (taps[4] - taps[5]) / delta], dim=-1)
    return grad, taps


def color_network(x, dirs, grad, feat, emb, p: Params, cfg: FieldCfg) -> torch.Tensor:
    """fields/sdf_field.py:532-612 get_colors. Note the RAW gradient enters (572-578).  The ref-nerf options: diffuse / tint heads
    (:536-540), reflected direction (:545-549), the shorter input list of use_diffuse_color (:566-571), n . v (:580-583), the
    combination and clamp (:596-607)."""
    if cfg.use_diffuse_color:
        raw_rgb_diffuse = F.linear(feat, p["diffuse_color_pred.weight"], p["diffuse_color_pred.bias"])
    if cfg.use_specular_tint:
        tint = torch.sigmoid(F.linear(feat, p["specular_tint_pred.weight"], p["specular_tint_pred.bias"]))
    normals = F.normalize(grad, p=2, dim=-1)
    if cfg.use_reflections:
        refdirs = 2.0 * torch.sum(normals * -dirs, dim=-1, keepdim=True) * normals + dirs
        d = nerf_encoding(refdirs, 4, include_input=True)
    else:
        d = nerf_encoding(dirs, 4, include_input=True)
    parts = [d, feat, emb] if cfg.use_diffuse_color else [x, d, grad, feat, emb]
    if cfg.use_n_dot_v:
        parts.append(torch.sum(normals * dirs, dim=-1, keepdim=True))
    h = torch.cat(parts, dim=-1)
    n_lin = cfg.num_layers_color + 1
    for l in range(n_lin):
        h = linear_wn(p, f"clin{l}", h)
        if l < n_lin - 1:
            z = h
            h = torch.relu(z)
            if RELU_HOOK is not None:  # test instrumentation, see relu_hook()
                if RELU_HOOK.get("record") is not None:
                    RELU_HOOK["record"][l] = z.detach()
                flip = (RELU_HOOK.get("flip") or {}).get(l)
                if flip is not None:  # the OTHER branch of the ReLU at the flagged (point, unit) pairs
                    h = torch.where(flip, z - h, h)
                force = (RELU_HOOK.get("force") or {}).get(l)
                if force is not None:  # a GIVEN branch pattern (True: the linear branch) whatever round-off makes of sign(z) in this run
                    h = torch.where(force, z, torch.zeros_like(z))
    rgb = torch.sigmoid(h)
    if cfg.use_diffuse_color:
        diffuse_linear = torch.sigmoid(raw_rgb_diffuse - math.log(3.0))
        specular_linear = tint * rgb if cfg.use_specular_tint else 0.5 * rgb
        rgb = torch.clamp(specular_linear + diffuse_linear, 0.0, 1.0)
    return rgb * (1 + 2 * cfg.rgb_padding) - cfg.rgb_padding


def laplace_density(sdf: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """fields/sdf_field.py:49-71. beta = |beta_param| + beta_min."""
    return (1.0 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def neus_inv_s(variance: torch.Tensor) -> torch.Tensor:
    """fields/sdf_field.py:116-118."""
    return torch.exp(variance * 10.0).clip(1e-6, 1e6)


def neus_alpha(sdf, grad, dirs, deltas, inv_s, cos_anneal_ratio: float) -> torch.Tensor:
    """fields/sdf_field.py:494-516.  sdf, deltas: [N,S]; grad: [N,S,3]; dirs: [N,3]."""
    true_cos = (dirs[:, None, :] * grad).sum(-1)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    est_next = sdf + iter_cos * deltas * 0.5
    est_prev = sdf - iter_cos * deltas * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


def field_outputs(origins, dirs, starts, deltas, cam_idx, p: Params, cfg: FieldCfg, mask=None,
                  cos_anneal_ratio: float = 1.0, training: bool = True, numerical_delta: Optional[float] = None,
                  contraction: str = "inf") -> Dict[str, torch.Tensor]:
    """fields/sdf_field.py:614-689 get_outputs(return_alphas=True) for a dense [N,S] sample set.  numerical_delta: the
    use_numerical_gradients branch (:638-644) with numerical_gradients_delta = that value."""
    n, s = starts.shape
    pos = origins[:, None, :] + dirs[:, None, :] * starts[..., None]  # cameras/rays.py:61-73 (START positions)
    x = (contract_inf if contraction == "inf" else contract_l2)(pos.reshape(-1, 3))  # sdf_field.py:629
    points_norm = x.norm(dim=-1)
    sampled_sdf = None
    if numerical_delta is None:
        sdf, feat, grad = sdf_and_gradient(x, p, cfg, mask)
    else:
        h = geo_network(x, p, cfg, mask)
        sdf, feat = h[:, 0], h[:, 1:]
        grad, taps = numerical_gradient(x, p, cfg, numerical_delta, mask)
        sampled_sdf = taps.view(6, n, s).permute(1, 2, 0).contiguous()  # :644
    dirs_flat = dirs[:, None, :].expand(n, s, 3).reshape(-1, 3)
    if training and cfg.use_appearance_embedding:
        emb = p["embedding_appearance.embedding.weight"][cam_idx][:, None, :].expand(n, s, -1).reshape(n * s, -1)
    else:
        emb = torch.zeros(n * s, cfg.appearance_embedding_dim, dtype=x.dtype)  # sdf_field.py:554-564
    rgb = color_network(x, dirs_flat, grad, feat, emb, p, cfg)
    beta = p["laplace_density.beta"].abs() + p["laplace_density.beta_min"]
    density = laplace_density(sdf, beta)
    inv_s = neus_inv_s(p["deviation_network.variance"])
    alpha = neus_alpha(sdf.view(n, s), grad.view(n, s, 3), dirs, deltas, inv_s, cos_anneal_ratio)
    return {
        "rgb": rgb.view(n, s, 3), "density": density.view(n, s), "sdf": sdf.view(n, s),
        "gradient": grad.view(n, s, 3), "normal": F.normalize(grad, p=2, dim=-1).view(n, s, 3),
        "points_norm": points_norm.view(n, s), "alpha": alpha, "geo_feature": feat.view(n, s, -1), "sampled_sdf": sampled_sdf,
    }


# ----------------------------------------------------------------------------- proposal density
def proposal_density(positions: torch.Tensor, p: Params, prefix: str, cfg: ProposalCfg, contraction: str = "inf") -> torch.Tensor:
    """fields/density_fields.py:99-118 (+ base_field.py:48-65): contraction, (x+2)/4, grid -> ReLU MLP -> exp."""
    shape = positions.shape[:-1]
    x = ((contract_inf if contraction == "inf" else contract_l2)(positions.reshape(-1, 3)) + 2.0) / 4.0
    lv = cfg.grid_levels()
    feat = hashgrid.grid_encode(x, p[f"{prefix}.table"].view(lv.n_entries, lv.n_features), lv)
    pre = torch.relu(feat @ p[f"{prefix}.w1"].t()) @ p[f"{prefix}.w2"].t()
    return trunc_exp(pre[:, 0]).view(shape)


class _TruncExp(torch.autograd.Function):
    """field_components/activations.py:23-39: exp forward, backward clamps the exponent to [-15,15]."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


# ----------------------------------------------------------------------------- weights + renderers
def weights_from_density(density: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """cameras/rays.py:146-167."""
    dd = deltas * density
    acc = torch.cumsum(dd[:, :-1], dim=-1)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[:, :1]), acc], dim=-1))
    return (1 - torch.exp(-dd)) * trans


def weights_from_alphas(alpha: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """cameras/rays.py:194-230: T = exclusive cumprod(1 - alpha + 1e-7). Returns weights [N,S], T [N,S+1]."""
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], dim=1), dim=1)
    return alpha * trans[:, :-1], trans


def render(weights, rgb, normals, starts, ends, background: Optional[torch.Tensor] = None):
    """model_components/renderers.py:81-92 (rgb), 196 (accumulation), 245-259 (expected depth), 294 (normals)."""
    acc = weights.sum(dim=1)
    out_rgb = (weights[..., None] * rgb).sum(dim=1)
    if background is not None:
        out_rgb = out_rgb + background * (1.0 - acc[:, None])
    steps = (starts + ends) / 2
    depth = (weights * steps).sum(dim=1) / (acc + 1e-10)
    depth = torch.clip(depth, steps.min(), steps.max())
    normal = (weights[..., None] * normals).sum(dim=1)
    return out_rgb, depth, normal, acc


# ----------------------------------------------------------------------------- samplers
def piecewise_spacing(x: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:240 spacing_fn of UniformLinDispPiecewiseSampler."""
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def piecewise_spacing_inv(x: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:241."""
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


def spacing_to_euclidean(bins: torch.Tensor, nears: torch.Tensor, fars: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:115-117."""
    s_near, s_far = piecewise_spacing(nears)[:, None], piecewise_spacing(fars)[:, None]
    return piecewise_spacing_inv(bins * s_far + (1 - bins) * s_near)


SPACINGS = {
    # spacing_fn / spacing_fn_inv of the SpacedSampler subclasses, ray_samplers.py:130-247
    "piecewise": (piecewise_spacing, piecewise_spacing_inv),
    "uniform": (lambda x: x, lambda x: x),
    "lindisp": (lambda x: 1 / x, lambda x: 1 / x),
    "sqrt": (torch.sqrt, lambda x: x ** 2),
    "log": (torch.log, torch.exp),
}


def spaced_to_euclidean(kind: str, bins: torch.Tensor, nears: torch.Tensor, fars: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:115-117 for any SpacedSampler subclass: fn_inv(x s_far + (1 - x) s_near)."""
    fn, inv = SPACINGS[kind]
    s_near, s_far = fn(nears)[:, None], fn(fars)[:, None]
    return inv(bins * s_far + (1 - bins) * s_near)


def initial_bins(n_rays: int, num_samples: int, t_rand: Optional[torch.Tensor], dtype=torch.float32):
    """ray_samplers.py:101-113. t_rand: [N,1] single-jitter draw, [N,S+1] per-bin-edge draws (single_jitter=False, :107-110), or None
    for eval-mode (deterministic) bins."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1, dtype=dtype)[None, :].expand(n_rays, -1)
    if t_rand is not None:
        centers = (bins[:, 1:] + bins[:, :-1]) / 2.0
        upper = torch.cat([centers, bins[:, -1:]], -1)
        lower = torch.cat([bins[:, :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    return bins


def pdf_sample(weights: torch.Tensor, existing_bins: torch.Tensor, num_samples: int,
               u_rand: Optional[torch.Tensor], histogram_padding: float = 0.01, eps: float = 1e-5) -> torch.Tensor:
    """ray_samplers.py:303-358, include_original=False.  weights [N,S_in], existing_bins [N,S_in+1] (spacing).

    u_rand: [N,1] uniform draw (single jitter) or None (eval: bin centres).  Returns new bins [N,num_samples+1].
    """
    num_bins = num_samples + 1
    w = weights + histogram_padding
    w_sum = w.sum(dim=-1, keepdim=True)
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, dtype=w.dtype)
    if u_rand is not None:
        u = u[None, :] + u_rand / num_bins
    else:
        u = (u + 1.0 / (2 * num_bins))[None, :].expand(w.shape[0], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, side="right")
    hi = existing_bins.shape[-1] - 1
    below = torch.clamp(inds - 1, 0, hi)
    above = torch.clamp(inds, 0, hi)
    cdf_g0, cdf_g1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b_g0, b_g1 = torch.gather(existing_bins, -1, below), torch.gather(existing_bins, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    return (b_g0 + t * (b_g1 - b_g0)).detach()


def proposal_sampler(origins, dirs, nears, fars, p: Params, cfg: ModelCfg, anneal: float,
                     rand: Optional[List[torch.Tensor]]):
    """ray_samplers.py:537-578 with update_sched == -1 (neus_facto.py:138): proposal nets always in the graph.

    rand: [t_rand0 [N,1], u_rand1 [N,1], u_rand2 [N,1]] for training-mode jitter, or None for eval-mode.
    Returns final (bins, starts, ends) and the per-level lists used by the interlevel loss.
    """
    n = origins.shape[0]
    weights_list, bins_list = [], []
    bins = initial_bins(n, cfg.num_proposal_samples[0], None if rand is None else rand[0], origins.dtype)
    weights = None
    n_prop = len(cfg.proposals)
    for lvl in range(n_prop + 1):
        if lvl > 0:
            ns = cfg.num_proposal_samples[lvl] if lvl < n_prop else cfg.num_neus_samples
            bins = pdf_sample(torch.pow(weights, anneal), bins, ns, None if rand is None else rand[lvl],
                              cfg.histogram_padding)
        eu = spacing_to_euclidean(bins, nears, fars)
        starts, ends = eu[:, :-1], eu[:, 1:]
        if lvl < n_prop:
            mid = origins[:, None, :] + dirs[:, None, :] * ((starts + ends) / 2)[..., None]  # rays.py:46-55 MIDPOINT
            dens = proposal_density(mid, p, f"proposal_networks.{lvl}", cfg.proposals[lvl])
            weights = weights_from_density(dens, ends - starts)
            weights_list.append(weights)
            bins_list.append(bins)
    return bins, starts, ends, weights_list, bins_list


# ----------------------------------------------------------------------------- NeuS hierarchical sampler
def uniform_to_euclidean(bins: torch.Tensor, nears: torch.Tensor, fars: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:115-117 with UniformSampler's identity spacing_fn (:130-151)."""
    return bins * fars[:, None] + (1 - bins) * nears[:, None]


def neus_upsample_alpha(sdf: torch.Tensor, deltas: torch.Tensor, inv_s: float) -> torch.Tensor:
    """ray_samplers.py:899-944 rendering_sdf_with_fixed_inv_s.  sdf [N,S], deltas [N,S] -> alpha [N,S-1]."""
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    d = deltas[:, :-1]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (d + 1e-5)
    prev_cos = torch.cat([torch.zeros_like(cos_val[:, :1]), cos_val[:, :-1]], dim=-1)
    cos_val = torch.minimum(prev_cos, cos_val).clip(-1e3, 0.0)
    prev_esti = mid_sdf - cos_val * d * 0.5
    next_esti = mid_sdf + cos_val * d * 0.5
    prev_cdf = torch.sigmoid(prev_esti * inv_s)
    next_cdf = torch.sigmoid(next_esti * inv_s)
    return (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)


def merge_bins(bins_1: torch.Tensor, bins_2: torch.Tensor):
    """ray_samplers.py:757-786 merge_ray_samples on spacing bins [N,S1+1], [N,S2+1] -> merged bins [N,S1+S2+1] and the
    sorted index into cat(starts_1, starts_2)."""
    ends = torch.maximum(bins_1[:, -1:], bins_2[:, -1:])
    merged, index = torch.sort(torch.cat([bins_1[:, :-1], bins_2[:, :-1]], -1), -1)
    return torch.cat([merged, ends], dim=-1), index


def unisurf_sampler(nears, fars, sdf_fn, occupancy_fn, delta: float, rand=None, num_samples_interval: int = 64,
                    num_samples_outside: int = 32, num_samples_importance: int = 32, num_marching_steps: int = 256):
    """UniSurfSampler.generate_ray_samples, ray_samplers.py:996-1093 (single_jitter=False, the default: one draw per bin edge).

    sdf_fn(starts [N,S]) -> sdf [N,S].  rand: the four torch.rand draws in call order - marching bins [N,M+1], importance
    resampling [N,K+1], outside bins [N,O+1], interval bins [N,I+1] - or None (eval).  Returns a dict with the merged euclidean
    bins [N,I+K+O+1], the surface mask / depth z and the pieces the tests look at."""
    n = nears.shape[0]
    r = rand if rand is not None else [None] * 4
    m_bins = initial_bins(n, num_marching_steps, r[0], nears.dtype)
    m_eu = uniform_to_euclidean(m_bins, nears, fars)
    starts = m_eu[:, :-1]
    with torch.no_grad():
        sdf = sdf_fn(starts)
    occ = occupancy_fn(sdf)
    weights, _ = weights_from_alphas(occ)                                   # :1016-1017
    imp_bins = pdf_sample(weights, m_bins, num_samples_importance, r[1], histogram_padding=1e-5)   # :1019-1024 (padding :982)
    out_bins = initial_bins(n, num_samples_outside, r[2], nears.dtype)       # :1027
    ui_bins, _ = merge_bins(imp_bins, out_bins)                             # :1030-1032 (spacing domain, uniform spacing)
    ui_eu = uniform_to_euclidean(ui_bins, nears, fars)
    # first outside -> inside sign change (:1037-1052)
    sgn = torch.cat([torch.sign(sdf[:, :-1] * sdf[:, 1:]), torch.ones(n, 1, dtype=sdf.dtype)], dim=-1)
    cost = sgn * torch.arange(sdf.shape[1], 0, -1, dtype=sdf.dtype)
    values, idx = torch.min(cost, -1)
    ar = torch.arange(n)
    mask = (values < 0) & (sdf[ar, idx] > 0)
    hi = torch.clamp(idx + 1, max=sdf.shape[1] - 1)
    d_low, v_low, d_high, v_high = starts[ar, idx], sdf[ar, idx], starts[ar, hi], sdf[ar, hi]
    z = (v_low * d_high - v_high * d_low) / (v_low - v_high)               # :1063 linear interpolation
    dists = fars - nears
    nn = torch.where(mask, torch.maximum(z - dists * delta, nears), nears)  # :1068-1075
    nf = torch.where(mask, torch.minimum(z + dists * delta, fars), fars)
    i_bins = initial_bins(n, num_samples_interval, r[3], nears.dtype)        # :1078
    i_eu = uniform_to_euclidean(i_bins, nn, nf)
    merged, _ = merge_bins(i_eu, ui_eu)                                     # merge_ray_samples_in_eculidean :1095-1130
    return {"bins": merged, "mask": mask, "z": z, "marching_starts": starts, "sdf": sdf, "importance_bins": imp_bins,
            "interval_bins": i_eu, "new_nears": nn, "new_fars": nf}


def neus_sampler(origins, dirs, nears, fars, sdf_fn, num_samples: int = 64, num_samples_importance: int = 64,
                 num_upsample_steps: int = 4, base_variance: float = 64.0, rand: Optional[List[torch.Tensor]] = None):
    """ray_samplers.py:815-897 NeuSSampler.generate_ray_samples.

    sdf_fn(starts [N,S]) -> sdf [N,S] evaluates the field at the frustum START positions (sdf_field.py:412-418).
    rand: [t_rand0, u_rand1, ..., u_rand_steps] single-jitter draws [N,1] (training) or None (eval).
    Returns the final spacing bins [N, num_samples + num_samples_importance + 1] and their euclidean starts / ends.
    """
    n = origins.shape[0]
    bins = initial_bins(n, num_samples, None if rand is None else rand[0], origins.dtype)
    n_new = num_samples_importance // num_upsample_steps
    sdf = None
    new_bins, index = bins, None
    for it in range(num_upsample_steps):
        eu_new = uniform_to_euclidean(new_bins, nears, fars)
        with torch.no_grad():
            new_sdf = sdf_fn(eu_new[:, :-1])
        if index is not None:
            sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, index)
        else:
            sdf = new_sdf
        eu = uniform_to_euclidean(bins, nears, fars)
        alphas = neus_upsample_alpha(sdf, eu[:, 1:] - eu[:, :-1], base_variance * 2**it)
        weights, _ = weights_from_alphas(alphas)
        weights = torch.cat([weights, torch.zeros_like(weights[:, :1])], dim=1)
        new_bins = pdf_sample(weights, bins, n_new, None if rand is None else rand[1 + it], histogram_padding=1e-5)
        bins, index = merge_bins(bins, new_bins)
    eu = uniform_to_euclidean(bins, nears, fars)
    return bins, eu[:, :-1], eu[:, 1:]


def neus_forward(origins, dirs, cam_idx, p: Params, cfg: "ModelCfg", cos_anneal_ratio: float = 1.0, rand=None, mask=None,
                 training=True, num_samples: int = 64, num_samples_importance: int = 64, num_upsample_steps: int = 4,
                 base_variance: float = 64.0, samples=None):
    """models/neus.py:94-104 + base_surface_model.py:292-365 (background_model == 'none', black background).
    samples = (bins, starts, ends) bypasses the sampler (parity of the field / renderer on identical samples: four rounds of
    inverse-CDF resampling with histogram_padding 1e-5 are ill-conditioned in fp32, see tests)."""
    n = origins.shape[0]
    nears = torch.full((n,), cfg.near, dtype=origins.dtype)
    fars = torch.full((n,), cfg.far, dtype=origins.dtype)

    def sdf_fn(starts):
        pos = origins[:, None, :] + dirs[:, None, :] * starts[..., None]  # NOT contracted (sdf_field.py:412-418)
        return geo_network(pos.reshape(-1, 3), p, cfg.field, mask)[:, 0].view(starts.shape)

    if samples is None:
        bins, starts, ends = neus_sampler(origins, dirs, nears, fars, sdf_fn, num_samples, num_samples_importance,
                                          num_upsample_steps, base_variance, rand)
    else:
        bins, starts, ends = samples
    fo = field_outputs(origins, dirs, starts, ends - starts, cam_idx, p, cfg.field, mask, cos_anneal_ratio, training)
    weights, trans = weights_from_alphas(fo["alpha"])
    rgb, depth, normal, acc = render(weights, fo["rgb"], fo["normal"], starts, ends)
    return {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": acc, "weights": weights, "field": fo,
            "starts": starts, "ends": ends, "bins": bins}


# ----------------------------------------------------------------------------- VolSDF error-bounded sampler
def volsdf_dstar(sdf: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:704-726 get_dstar (Theorem 1 of VolSDF).  sdf, deltas [N,S] -> d* [N,S] (last column repeated)."""
    a, b, c = deltas[:, :-1], sdf[:, :-1].abs(), sdf[:, 1:].abs()
    first = a.pow(2) + b.pow(2) <= c.pow(2)
    second = a.pow(2) + c.pow(2) <= b.pow(2)
    d_star = torch.zeros_like(a)
    d_star = torch.where(first, b, d_star)
    d_star = torch.where(second, c, d_star)
    s = (a + b + c) / 2.0
    area = s * (s - a) * (s - b) * (s - c)
    mask = ~first & ~second & (b + c - a > 0)
    d_star = torch.where(mask, (2.0 * torch.sqrt(area)) / a, d_star)
    d_star = (sdf[:, 1:].sign() * sdf[:, :-1].sign() == 1) * d_star
    return torch.cat((d_star, d_star[:, -1:]), dim=-1)


def volsdf_error_bound(beta: torch.Tensor, sdf: torch.Tensor, d_star: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """ray_samplers.py:740-755 get_error_bound.  beta [N,1] (or [1]); returns [N]."""
    dd = deltas * laplace_density(sdf, beta)
    integral = torch.cat([torch.zeros_like(dd[:, :1]), torch.cumsum(dd[:, :-1], dim=-1)], dim=-1)
    per_section = torch.exp(-d_star / beta) * (deltas ** 2.0) / (4 * beta ** 2)
    err_int = torch.cumsum(per_section, dim=-1)
    bound = (torch.clamp(torch.exp(err_int), max=1.0e6) - 1.0) * torch.exp(-integral)
    return bound.max(-1)[0]


def volsdf_update_beta(beta0: torch.Tensor, beta: torch.Tensor, sdf, d_star, deltas, eps: float = 0.1, iters: int = 10):
    """ray_samplers.py:728-738 get_updated_beta (bisection; the reference's in-place aliasing written functionally)."""
    curr = volsdf_error_bound(beta0, sdf, d_star, deltas)
    beta_max = torch.where(curr <= eps, beta0.expand_as(beta), beta)
    beta_min = beta0.expand_as(beta).clone()
    for _ in range(iters):
        mid = (beta_min + beta_max) / 2.0
        e = volsdf_error_bound(mid[:, None], sdf, d_star, deltas)
        beta_max = torch.where(e <= eps, mid, beta_max)
        beta_min = torch.where(e > eps, mid, beta_min)
    return beta_max


def weights_and_transmittance_from_density(density: torch.Tensor, deltas: torch.Tensor):
    """cameras/rays.py:169-192: weights [N,S] and transmittance [N,S] (T_i BEFORE sample i)."""
    dd = deltas * density
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[:, :1]), torch.cumsum(dd[:, :-1], dim=-1)], dim=-1))
    return (1 - torch.exp(-dd)) * trans, trans


def error_bounded_sampler(nears, fars, sdf_fn, beta0: torch.Tensor, num_samples: int = 64, num_samples_eval: int = 128,
                          num_samples_extra: int = 32, eps: float = 0.1, beta_iters: int = 10, max_total_iters: int = 5,
                          rand: Optional[List[torch.Tensor]] = None, trace: Optional[list] = None):
    """ray_samplers.py:611-702 ErrorBoundedSampler.generate_ray_samples (VolSDF Algorithm 1), UniformSampler spacing.

    sdf_fn(starts [N,S]) -> sdf [N,S].  beta0 = density_fn.get_beta() (shape [1]).  rand: the torch.rand draws in call order
    (initial uniform [N,S_eval+1], one [N,n+1] per PDF call, extra uniform [N,extra+1]) or None (eval).
    Returns final spacing bins and euclidean starts / ends ([N, num_samples + num_samples_extra])."""
    n = nears.shape[0]
    rq = list(rand) if rand is not None else None
    draw = (lambda: rq.pop(0)) if rq is not None else (lambda: None)
    bins = initial_bins(n, num_samples_eval, draw(), nears.dtype)
    eu = uniform_to_euclidean(bins, nears, fars)
    deltas = eu[:, 1:] - eu[:, :-1]
    beta = torch.sqrt((1.0 / (4.0 * math.log(eps + 1.0))) * (deltas ** 2.0).sum(-1))
    total, not_converge, index, sdf, new_bins = 0, True, None, None, bins
    while not_converge and total < max_total_iters:
        with torch.no_grad():
            new_sdf = sdf_fn(uniform_to_euclidean(new_bins, nears, fars)[:, :-1])
        sdf = new_sdf if index is None else torch.gather(torch.cat([sdf, new_sdf], -1), 1, index)
        eu = uniform_to_euclidean(bins, nears, fars)
        deltas = eu[:, 1:] - eu[:, :-1]
        d_star = volsdf_dstar(sdf, deltas)
        rec = {"bins_in": bins, "sdf_in": sdf, "beta_in": beta}
        beta = volsdf_update_beta(beta0, beta, sdf, d_star, deltas, eps, beta_iters)
        weights, trans = weights_and_transmittance_from_density(laplace_density(sdf, beta[:, None]), deltas)
        total += 1
        not_converge = bool(beta.max() > beta0)
        rec.update({"beta_out": beta, "weights": weights})
        if not_converge and total < max_total_iters:
            per_section = torch.exp(-d_star / beta[:, None]) * (deltas ** 2.0) / (4 * beta[:, None] ** 2)
            w = (torch.clamp(torch.exp(torch.cumsum(per_section, dim=-1)), max=1.0e6) - 1.0) * trans
            new_bins = pdf_sample(w, bins, num_samples_eval, draw(), histogram_padding=1e-5)
            rec.update({"err_weights": w, "new_bins": new_bins})
            bins, index = merge_bins(bins, new_bins)
            rec.update({"merged_bins": bins, "index": index})
        else:
            bins = pdf_sample(weights, bins, num_samples, draw(), histogram_padding=1e-5)
            rec.update({"final_bins": bins})
        if trace is not None:
            trace.append(rec)
    if num_samples_extra > 0:
        bins, _ = merge_bins(bins, initial_bins(n, num_samples_extra, draw(), nears.dtype))
    eu = uniform_to_euclidean(bins, nears, fars)
    return bins, eu[:, :-1], eu[:, 1:]


def volsdf_forward(origins, dirs, cam_idx, p: Params, cfg: "ModelCfg", rand=None, mask=None, training=True,
                   num_samples: int = 64, num_samples_eval: int = 128, num_samples_extra: int = 32, samples=None, trace=None):
    """models/volsdf.py:62-79 + base_surface_model.py:292-365 (background_model == 'none', black background): density from the
    Laplace CDF of the sdf (sdf_field.py:49-71), weights from density (rays.py:146-167)."""
    n = origins.shape[0]
    nears = torch.full((n,), cfg.near, dtype=origins.dtype)
    fars = torch.full((n,), cfg.far, dtype=origins.dtype)

    def sdf_fn(starts):
        pos = origins[:, None, :] + dirs[:, None, :] * starts[..., None]  # NOT contracted (sdf_field.py:412-418)
        return geo_network(pos.reshape(-1, 3), p, cfg.field, mask)[:, 0].view(starts.shape)

    if samples is None:
        beta0 = (p["laplace_density.beta"].abs() + p["laplace_density.beta_min"]).detach()
        bins, starts, ends = error_bounded_sampler(nears, fars, sdf_fn, beta0, num_samples, num_samples_eval, num_samples_extra,
                                                   rand=rand, trace=trace)
    else:
        bins, starts, ends = samples
    fo = field_outputs(origins, dirs, starts, ends - starts, cam_idx, p, cfg.field, mask, 1.0, training)
    weights, trans = weights_and_transmittance_from_density(fo["density"], ends - starts)
    rgb, depth, normal, acc = render(weights, fo["rgb"], fo["normal"], starts, ends)
    return {"rgb": rgb, "depth": depth, "normal": normal, "accumulation": acc, "weights": weights, "field": fo,
            "starts": starts, "ends": ends, "bins": bins}


# ----------------------------------------------------------------------------- losses
def _blur_stepfun(x, y, r):
    """model_components/losses.py:116-128."""
    xc = torch.cat([x - r, x + r], dim=-1)
    xr, idx = torch.sort(xc, dim=-1)
    zeros = torch.zeros_like(y[:, :1])
    y1 = (torch.cat([y, zeros], dim=-1) - torch.cat([zeros, y], dim=-1)) / (2 * r)
    y2 = torch.gather(torch.cat([y1, -y1], dim=-1), -1, idx[:, :-1])
    yr = torch.cumsum((xr[:, 1:] - xr[:, :-1]) * torch.cumsum(y2, dim=-1), dim=-1)
    return xr, torch.cat([zeros, yr], dim=-1)


def interlevel_loss_zip(weights_list: List[torch.Tensor], bins_list: List[torch.Tensor]) -> torch.Tensor:
    """model_components/losses.py:131-172. Last entries are the (detached) field weights / bins."""
    c = bins_list[-1].detach()
    w = weights_list[-1].detach()
    wn = w / (c[:, 1:] - c[:, :-1])
    loss = 0.0
    for cp, wp, r in zip(bins_list[:-1], weights_list[:-1], [0.03, 0.003]):
        xr, yr = _blur_stepfun(c, wn, r)
        yr = torch.clip(yr, min=0)
        ycum = torch.cumsum((yr[:, 1:] + yr[:, :-1]) * 0.5 * (xr[:, 1:] - xr[:, :-1]), dim=-1)
        ycum = torch.cat([torch.zeros_like(ycum[:, :1]), ycum], dim=-1)
        inds = torch.searchsorted(xr, cp.contiguous(), side="right")
        below = torch.clamp(inds - 1, 0, xr.shape[-1] - 1)
        above = torch.clamp(inds, 0, xr.shape[-1] - 1)
        x0, x1 = torch.gather(xr, -1, below), torch.gather(xr, -1, above)
        y0, y1 = torch.gather(ycum, -1, below), torch.gather(ycum, -1, above)
        t = torch.clip(torch.nan_to_num((cp - x0) / (x1 - x0), 0), 0, 1)
        b = y0 + t * (y1 - y0)
        w_gt = b[:, 1:] - b[:, :-1]
        loss = loss + torch.mean(torch.clip(w_gt - wp, min=0) ** 2 / (wp + 1e-5))
    return loss


# ----------------------------------------------------------------------------- full model step
def neus_facto_forward(origins, dirs, cam_idx, p: Params, cfg: ModelCfg, anneal: float = 1.0,
                       cos_anneal_ratio: float = 1.0, rand=None, mask=None, training=True, nears=None, fars=None,
                       numerical_delta: Optional[float] = None, background: Optional[Dict] = None):
    """models/neus_facto.py:282-302 + base_surface_model.py:292-365, black bg.  nears / fars [N]:
    per-ray planes from a box / sphere collider (scene_colliders.py:47-109,132-170); default: the NearFarCollider's constants.
    numerical_delta: the field's use_numerical_gradients branch (sdf_field.py:638-644; neus-facto-angelo, BASELINE config 5).
    background: None (background_model == 'none'), or {"prefix": "field_background.", "lv": GridLevels, "geo_feat_dim": 15} for
    background_model == 'grid' (neus_facto.py:289-290 -> forward_background_field_and_merge, base_surface_model.py:266-290: the
    background field is evaluated on the SDF samples themselves and replaces alpha and colour of the samples whose START position
    lies outside the unit sphere; a NeuS-facto model has no 'bg_transmittance', so :314-329 never runs for it)."""
    n = origins.shape[0]
    if nears is None:
        nears = torch.full((n,), cfg.near, dtype=origins.dtype)  # scene_colliders.py:124-129
        fars = torch.full((n,), cfg.far, dtype=origins.dtype)
    bins, starts, ends, weights_list, bins_list = proposal_sampler(origins, dirs, nears, fars, p, cfg, anneal, rand)
    deltas = ends - starts
    fo = field_outputs(origins, dirs, starts, deltas, cam_idx, p, cfg.field, mask, cos_anneal_ratio, training, numerical_delta=numerical_delta)
    alpha, rgb_s = fo["alpha"], fo["rgb"]
    if background is not None:
        pos = origins[:, None, :] + dirs[:, None, :] * starts[..., None]
        inside = (pos.norm(dim=-1) < 1.0).to(alpha.dtype)  # get_foreground_mask, base_surface_model.py:256-264
        bg = nerfacto_field(origins, dirs, starts, ends, cam_idx, p, background["prefix"], background["lv"],
                            geo_feat_dim=background.get("geo_feat_dim", 15), training=training)
        bg_alpha = 1.0 - torch.exp(-deltas * bg["density"])  # RaySamples.get_alphas, cameras/rays.py:131-144
        alpha = alpha * inside + (1.0 - inside) * bg_alpha
        rgb_s = rgb_s * inside[..., None] + (1.0 - inside[..., None]) * bg["rgb"]
        fo = dict(fo, alpha=alpha, rgb=rgb_s, inside=inside)
    weights, trans = weights_from_alphas(alpha)
    rgb, depth, normal, acc = render(weights, rgb_s, fo["normal"], starts, ends)
    return {
        "rgb": rgb, "depth": depth, "normal": normal, "accumulation": acc, "weights": weights,
        "field": fo, "starts": starts, "ends": ends, "bins": bins,
        "weights_list": weights_list + [weights], "bins_list": bins_list + [bins],
    }


def neus_facto_loss(out, image, cfg: ModelCfg, curvature: Optional[Tuple[float, float]] = None) -> Dict[str, torch.Tensor]:
    """base_surface_model.py:399-406 (L1 rgb, eikonal) + neus_facto.py:304-310 (interlevel).  curvature = (delta, multiplier incl. the
    schedule's factor): neus_facto.py:312-325 on the six tap values of the numerical-gradient field."""
    g = out["field"]["gradient"]
    loss = {
        "rgb_loss": F.l1_loss(out["rgb"], image),
        "eikonal_loss": ((g.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult,
        "interlevel_loss": cfg.interlevel_loss_mult * interlevel_loss_zip(out["weights_list"], out["bins_list"]),
    }
    if curvature is not None:
        delta, mult = curvature
        sdf, taps = out["field"]["sdf"], out["field"]["sampled_sdf"]
        curv = (taps.reshape(sdf.shape + (3, 2)).sum(dim=-1) - 2 * sdf[..., None]) / (delta * delta)
        a = curv.abs()
        if RELU_HOOK is not None:  # test instrumentation (relu_hook): |c| is the path's other knife edge - the second difference divides
            # the sdf's round-off by delta^2, so two fp32 evaluations may disagree on sign(c) where |c| is below that noise
            if RELU_HOOK.get("record") is not None:
                RELU_HOOK["record"][-1] = curv.detach().reshape(-1, 3)
            flip = (RELU_HOOK.get("flip") or {}).get(-1)
            if flip is not None:  # the OTHER branch of |c| at the flagged elements
                a = torch.where(flip.view_as(a), -a, a)
            force = (RELU_HOOK.get("force") or {}).get(-1)
            if force is not None:  # a GIVEN branch pattern (True: +c)
                a = torch.where(force.view_as(a), curv, -curv)
        loss["curvature_loss"] = a.mean() * mult
    return loss


def monosdf_normal_loss(normal_pred: torch.Tensor, normal_gt: torch.Tensor) -> torch.Tensor:
    """model_components/losses.py:264-275 (checker of the fused surface_losses operator; pinned on the reference's own function by the CPU suite): L1 + cosine between the rendered normal and the monocular normal prior."""
    n_gt = F.normalize(normal_gt, p=2, dim=-1)
    n_pr = F.normalize(normal_pred, p=2, dim=-1)
    return torch.abs(n_pr - n_gt).sum(dim=-1).mean() + (1.0 - torch.sum(n_pr * n_gt, dim=-1)).mean()



# ----------------------------------------------------------------------------- parameter init
def init_field_params(cfg: FieldCfg, num_images: int = 49, seed: int = 0, dtype=torch.float32) -> Params:
    """Geometric init of fields/sdf_field.py:286-313, colour init :354-363, grid U(-1e-4,1e-4) (tcnn default)."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    dims = cfg.geo_dims()
    n_lin = len(dims) - 1
    for l in range(n_lin):
        out_dim = dims[l + 1] - dims[0] if (l + 1) in cfg.skip_in else dims[l + 1]
        w = torch.zeros(out_dim, dims[l])
        b = torch.zeros(out_dim)
        if l == n_lin - 1:
            sign = -1.0 if cfg.inside_outside else 1.0
            w.normal_(mean=sign * math.sqrt(math.pi) / math.sqrt(dims[l]), std=0.0001, generator=g)
            b.fill_(-sign * cfg.bias)
        elif l == 0:
            w[:, :3].normal_(0.0, math.sqrt(2) / math.sqrt(out_dim), generator=g)
        elif l in cfg.skip_in:
            w.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim), generator=g)
            w[:, -(dims[0] - 3):] = 0.0
        else:
            w.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim), generator=g)
        p[f"glin{l}.weight_v"] = w
        p[f"glin{l}.weight_g"] = w.norm(dim=1, keepdim=True)
        p[f"glin{l}.bias"] = b
    cd = cfg.color_dims()
    for l in range(len(cd) - 1):
        bound = math.sqrt(6.0 / cd[l])  # kaiming_uniform_, a=0, fan_in
        w = (torch.rand(cd[l + 1], cd[l], generator=g) * 2 - 1) * bound
        p[f"clin{l}.weight_v"] = w
        p[f"clin{l}.weight_g"] = w.norm(dim=1, keepdim=True)
        p[f"clin{l}.bias"] = torch.zeros(cd[l + 1])
    for name, on in (("diffuse_color_pred", cfg.use_diffuse_color), ("specular_tint_pred", cfg.use_specular_tint)):
        if on:  # sdf_field.py:333-336: nn.Linear(geo_feat_dim, 3), torch's default init (kaiming_uniform(a = sqrt 5) = U(+-1 / sqrt(fan_in)))
            bound = 1.0 / math.sqrt(cfg.geo_feat_dim)
            p[f"{name}.weight"] = (torch.rand(3, cfg.geo_feat_dim, generator=g) * 2 - 1) * bound
            p[f"{name}.bias"] = (torch.rand(3, generator=g) * 2 - 1) * bound
    lv = cfg.grid_levels()
    p["encoding.params"] = (torch.rand(lv.n_params, generator=g) * 2 - 1) * 1e-4
    p["laplace_density.beta"] = torch.full((1,), cfg.beta_init)
    p["laplace_density.beta_min"] = torch.full((1,), 1e-4)
    p["deviation_network.variance"] = torch.full((1,), cfg.beta_init)
    p["embedding_appearance.embedding.weight"] = torch.randn(num_images, cfg.appearance_embedding_dim, generator=g)
    return {k: v.to(dtype) for k, v in p.items()}


def init_proposal_params(cfgs, seed: int = 1, dtype=torch.float32) -> Params:
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for i, c in enumerate(cfgs):
        lv = c.grid_levels()
        d_in = lv.n_output_dims
        p[f"proposal_networks.{i}.table"] = (torch.rand(lv.n_params, generator=g) * 2 - 1) * 1e-4
        a1 = math.sqrt(6.0 / (d_in + c.hidden_dim))
        p[f"proposal_networks.{i}.w1"] = (torch.rand(c.hidden_dim, d_in, generator=g) * 2 - 1) * a1
        a2 = math.sqrt(6.0 / (c.hidden_dim + 1))
        p[f"proposal_networks.{i}.w2"] = (torch.rand(1, c.hidden_dim, generator=g) * 2 - 1) * a2
    return {k: v.to(dtype) for k, v in p.items()}


def synthetic_rays(n: int, seed: int = 42, radius: float = 2.73, dtype=torch.float32):
    """cameras on a sphere of radius ~2.73 looking at the origin, pinhole jitter."""
    g = torch.Generator().manual_seed(seed)
    cam = torch.randint(0, 49, (n,), generator=g)
    # 49 camera centres on the upper hemisphere
    k = torch.arange(49, dtype=torch.float64)
    phi = k * 2.399963229728653  # golden angle
    z = 0.15 + 0.7 * (k + 0.5) / 49
    r = torch.sqrt(1 - z * z)
    centers = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], dim=-1) * radius
    o = centers[cam]
    target = (torch.rand(n, 3, generator=g, dtype=torch.float64) * 2 - 1) * 0.6
    d = target - o
    d = d / d.norm(dim=-1, keepdim=True)
    return o.to(dtype), d.to(dtype), cam


# ----------------------------------------------------------------------------- packed-sample path (NeuS-acc)
# The reference calls three operators of nerfacc == 0.3.5 (pyproject.toml:31), a CUDA-only dependency that is NOT vendored in
# /root/reference and cannot be built here: PARITY UNPINNED for ray_marching (restated from the published algorithm of
# nerfacc/cuda/csrc/ray_marching.cu + helpers_*.h of that release); render_weight_from_alpha and accumulate_along_rays are
# pinned by their definitions (exclusive product scan / index_add), checked against plain torch below.
def ray_marching(origins, dirs, t_min, t_max, roi_aabb, binary, step_size: float):
    """nerfacc.cuda.ray_marching(origins, dirs, t_min, t_max, roi, binary, ContractionType.AABB, step_size, cone_angle = 0) as
    called at model_components/ray_samplers.py:1474-1484.  fp32 arithmetic operation by operation (numpy scalars), positions
    with one rounding (fma).  Returns (packed_info [N,2] = (offset, count), ray_indices [P], t_starts [P,1], t_ends [P,1])."""
    import numpy as np

    f = np.float32
    o_, d_ = origins.numpy().astype(np.float32), dirs.numpy().astype(np.float32)
    tmin, tmax = t_min.numpy().astype(np.float32), t_max.numpy().astype(np.float32)
    roi = roi_aabb.numpy().astype(np.float32).reshape(2, 3)
    occ = binary.numpy().astype(bool)
    R = occ.shape[0]
    Rf, step, half = f(R), f(step_size), f(0.5)

    def fma(a, b, c):
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    def occupied(p):
        if any(p[k] < roi[0, k] or p[k] > roi[1, k] for k in range(3)):
            return False
        idx = []
        for k in range(3):
            u = f(f(f(p[k] - roi[0, k]) / f(roi[1, k] - roi[0, k])) * Rf)
            idx.append(min(max(int(u), 0), R - 1))
        return bool(occ[idx[0], idx[1], idx[2]])

    def advance(t, p, d, inv_d):
        tt = f(3.0e38)
        for k in range(3):
            ext = f(roi[1, k] - roi[0, k])
            u = f(f(f(p[k] - roi[0, k]) / ext) * Rf)
            sg = f(1.0) if d[k] > 0 else (f(-1.0) if d[k] < 0 else f(0.0))
            tk = f(f(f(f(np.floor(f(f(u + half) + f(half * sg))) - u) * inv_d[k]) / Rf) * ext)
            tt = min(tt, tk)
        target = f(t + max(tt, f(0.0)))
        while True:
            t = f(t + step)
            if not t < target:
                return t

    ray_idx, ts, te, info = [], [], [], []
    with np.errstate(divide="ignore"):
        for r in range(o_.shape[0]):
            o, d = o_[r], d_[r]
            inv_d = f(1.0) / d
            t0 = tmin[r]
            t1 = f(t0 + step)
            tm = f(f(t0 + t1) * half)
            n0 = len(ts)
            while tm < tmax[r]:
                p = [fma(tm, d[k], o[k]) for k in range(3)]
                if occupied(p):
                    ray_idx.append(r)
                    ts.append(t0)
                    te.append(t1)
                    t0 = t1
                    t1 = f(t0 + step)
                    tm = f(f(t0 + t1) * half)
                else:
                    tm = advance(tm, p, d, inv_d)
                    t0 = f(tm - f(step * half))
                    t1 = f(tm + f(step * half))
            info.append((n0, len(ts) - n0))
    return (torch.tensor(info, dtype=torch.int64).view(-1, 2), torch.tensor(ray_idx, dtype=torch.int64),
            torch.tensor(np.array(ts, dtype=np.float32)).view(-1, 1), torch.tensor(np.array(te, dtype=np.float32)).view(-1, 1))


def ray_resampling(packed_info: torch.Tensor, t_starts: torch.Tensor, t_ends: torch.Tensor, weights: torch.Tensor, n_samples: int):
    """nerfacc.ray_resampling(packed_info, t_starts, t_ends, weights, n_samples) (model_components/ray_samplers.py:1496-1498), restated
    from nerfacc 0.3.5's pdf.cu (absent: PARITY UNPINNED): per ray with samples, n_samples + 1 edges at the inverse CDF of the padded,
    normalised weights, taken at u_j = 1 / (2 (n + 1)) + j (1 - 1 / (n + 1)) / n, fp32 operation by operation.
    Returns (packed_info [N,2], t_starts [P',1], t_ends [P',1])."""
    import numpy as np

    f = np.float32
    st, en, w = (x.reshape(-1).numpy().astype(np.float32) for x in (t_starts, t_ends, weights))
    info, os_, oe_ = [], [], []
    nb = n_samples + 1
    for off, cnt in packed_info.tolist():
        if cnt == 0:
            info.append((len(os_), 0))
            continue
        info.append((len(os_), n_samples))
        ws = f(0.0)
        for j in range(cnt):
            ws = f(ws + w[off + j])
        padding = max(f(f(1e-5) - ws), f(0.0))
        pad_step = f(padding / f(cnt))
        ws = f(ws + padding)
        step = f(f(f(1.0) - f(f(1.0) / f(nb))) / f(n_samples))
        idx, j = 0, 0
        prev, nxt = f(0.0), f(f(w[off] + pad_step) / ws)
        u = f(f(1.0) / f(2 * nb))
        starts, ends = [None] * n_samples, [None] * n_samples
        while j < nb:
            if u < nxt or idx == cnt - 1:
                scaling = f(f(en[off + idx] - st[off + idx]) / f(nxt - prev))
                t = f(np.float64(f(u - prev)) * np.float64(scaling) + np.float64(st[off + idx]))  # fused multiply-add
                if j < nb - 1:
                    starts[j] = t
                if j > 0:
                    ends[j - 1] = t
                u = f(u + step)
                j += 1
            else:
                idx += 1
                prev = nxt
                nxt = f(nxt + f(f(w[off + idx] + pad_step) / ws))
        os_ += starts
        oe_ += ends
    return (torch.tensor(info, dtype=torch.int64).view(-1, 2), torch.tensor(np.array(os_, dtype=np.float32)).view(-1, 1),
            torch.tensor(np.array(oe_, dtype=np.float32)).view(-1, 1))


def packed_weights_from_alpha(alpha: torch.Tensor, packed_info: torch.Tensor) -> torch.Tensor:
    """nerfacc.render_weight_from_alpha (models/neus_acc.py:103-107): w_i = alpha_i prod_{j<i, same ray} (1 - alpha_j).  alpha [P]."""
    out = []
    for off, cnt in packed_info.tolist():
        a = alpha[off:off + cnt]
        T = torch.cumprod(torch.cat([torch.ones_like(a[:1]), 1.0 - a[:-1]]), dim=0) if cnt > 0 else a
        out.append(a * T)
    return torch.cat(out) if out else alpha


def accumulate_along_rays(weights: torch.Tensor, ray_indices: torch.Tensor, values: Optional[torch.Tensor], n_rays: int) -> torch.Tensor:
    """nerfacc.accumulate_along_rays (models/neus_acc.py:108-121): index_add of weights * values (values None: of the weights)."""
    src = weights[:, None] * values if values is not None else weights[:, None]
    return torch.zeros(n_rays, src.shape[-1], dtype=src.dtype).index_add_(0, ray_indices, src)


def neus_acc_binary_update(binary: torch.Tensor, cube_coordinate: torch.Tensor, sdf_fn, inv_s: torch.Tensor, voxel_size: float,
                           step_size: float, alpha_thres: float = 0.001) -> torch.Tensor:
    """NeuSAccSampler.update_binary_grid (model_components/ray_samplers.py:1383-1432): occupied voxels whose |sdf| minus the voxel's
    half diagonal still yields an alpha above the threshold stay occupied (pruned voxels never come back)."""
    mask = binary.reshape(-1).clone()
    sdf = sdf_fn(cube_coordinate[mask]).abs()
    bound = voxel_size * (3 ** 0.5) / 2.0
    sdf = torch.maximum(sdf - bound, torch.zeros_like(sdf))
    prev_cdf = torch.sigmoid((sdf + step_size * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf - step_size * 0.5) * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    mask[mask.clone()] = alpha > alpha_thres
    return mask.reshape(binary.shape)


# ----------------------------------------------------------------------------- "grid" background field (BASELINE config 5)
def sh_degree4(d01: torch.Tensor) -> torch.Tensor:
    """tiny-cuda-nn's SphericalHarmonics encoding of degree 4 (spherical_harmonics.h of NVlabs/tiny-cuda-nn; the dependency is absent
    from /root/reference: PARITY UNPINNED, like the hash grid).  Input in [0,1]^3, mapped to [-1,1]^3 inside."""
    v = d01 * 2.0 - 1.0
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


def nerfacto_field(origins, dirs, starts, ends, cam_idx, p: Params, prefix: str, lv: "hashgrid.GridLevels", geo_feat_dim: int = 15,
                   training: bool = True, contraction: str = "inf") -> Dict[str, torch.Tensor]:
    """TCNNNerfactoField.forward (fields/nerfacto_field.py:225-330) as base_surface_model.py:181-187 builds it: contracted frustum
    MID points -> (x + 2) / 4 -> hash grid -> ReLU MLP without biases -> trunc_exp density + features; SH(view dir) | features |
    appearance embedding -> ReLU MLP -> sigmoid rgb.  Parameters under `prefix`: mlp_base.{table, w1, w2}, mlp_head.{w1, w2, w3},
    embedding_appearance.embedding.weight."""
    n, s = starts.shape
    mid = (starts + ends) / 2
    pos = origins[:, None, :] + dirs[:, None, :] * mid[..., None]
    x = (contract_inf if contraction == "inf" else contract_l2)(pos.reshape(-1, 3))
    x = (x + 2.0) / 4.0
    table = p[f"{prefix}mlp_base.table"].view(lv.n_entries, lv.n_features)
    feat = hashgrid.grid_encode(x, table, lv)
    h = torch.relu(feat @ p[f"{prefix}mlp_base.w1"].t()) @ p[f"{prefix}mlp_base.w2"].t()
    density = trunc_exp(h[:, :1])
    d = sh_degree4(((dirs + 1.0) / 2.0)[:, None, :].expand(n, s, 3).reshape(-1, 3))
    dim = p[f"{prefix}embedding_appearance.embedding.weight"].shape[1]
    if training:
        emb = p[f"{prefix}embedding_appearance.embedding.weight"][cam_idx][:, None, :].expand(n, s, -1).reshape(n * s, -1)
    else:
        emb = torch.zeros(n * s, dim, dtype=x.dtype)
    hh = torch.cat([d, h[:, 1:1 + geo_feat_dim], emb], dim=-1)
    rgb = torch.sigmoid(torch.relu(torch.relu(hh @ p[f"{prefix}mlp_head.w1"].t()) @ p[f"{prefix}mlp_head.w2"].t()) @ p[f"{prefix}mlp_head.w3"].t())
    return {"density": density.view(n, s), "rgb": rgb.view(n, s, 3)}


# ----------------------------------------------------------------------------- "mlp" background field (the reference's default)
def nerf_field(origins, dirs, starts, ends, p: Params, prefix: str = "", contraction: Optional[str] = "inf", skip: int = 4) -> Dict[str, torch.Tensor]:
    """NeRFField.forward (fields/vanilla_nerf_field.py:91-114, fields/base_field.py:111-126) as base_surface_model.py:189-200 builds
    it: contracted frustum MID points -> NeRFEncoding (10 frequencies, input appended; encodings.py:167-208) -> ReLU MLP whose layer
    `skip` takes cat([encoding, x]) (field_components/mlp.py:76-100, ReLU after every layer incl. the last) -> Softplus density head
    (field_heads.py:99-107); cat([NeRFEncoding(view dir, 4 frequencies, input appended), base output]) -> ReLU MLP -> Sigmoid rgb head.
    Parameters under `prefix` with the reference's state_dict names: mlp_base.layers.N.{weight,bias}, mlp_head.layers.N.*,
    field_output_density.net.*, field_heads.0.net.*."""
    n, s = starts.shape
    mid = (starts + ends) / 2
    pos = (origins[:, None, :] + dirs[:, None, :] * mid[..., None]).reshape(-1, 3)
    x = pos if contraction is None else (contract_inf if contraction == "inf" else contract_l2)(pos)

    def enc(v, nf):
        freqs = 2.0 ** torch.arange(nf, dtype=v.dtype)
        sc = (v[..., None] * freqs).reshape(v.shape[0], -1)
        return torch.cat([torch.sin(torch.cat([sc, sc + torch.pi / 2.0], dim=-1)), v], dim=-1)

    def mlp(name, inp, skip_at):
        h, l = inp, 0
        while f"{prefix}{name}.layers.{l}.weight" in p:
            if l == skip_at:
                h = torch.cat([inp, h], dim=-1)
            h = torch.relu(F.linear(h, p[f"{prefix}{name}.layers.{l}.weight"], p[f"{prefix}{name}.layers.{l}.bias"]))
            l += 1
        return h

    base = mlp("mlp_base", enc(x, 10), skip)
    density = F.softplus(F.linear(base, p[f"{prefix}field_output_density.net.weight"], p[f"{prefix}field_output_density.net.bias"]))
    d = dirs[:, None, :].expand(n, s, 3).reshape(-1, 3)
    head = mlp("mlp_head", torch.cat([enc(d, 4), base], dim=-1), -1)
    rgb = torch.sigmoid(F.linear(head, p[f"{prefix}field_heads.0.net.weight"], p[f"{prefix}field_heads.0.net.bias"]))
    return {"density": density.view(n, s), "rgb": rgb.view(n, s, 3)}


# ----------------------------------------------------------------------------- optimiser
def adam_reference(p, g, m, v, lr, beta1, beta2, eps, step, weight_decay=0.0, grad_scale=1.0):
    """torch.optim.Adam's single-tensor update (torch/optim/adam.py, the optimiser engine/optimizers.py:93-160 instantiates with
    eps 1e-15) on plain tensors, in place: the statement the adam_kernel restates; grad_scale = 1 / world folds the mean of the
    gradient all-reduce into the step."""
    g = g * grad_scale
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
