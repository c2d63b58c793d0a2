"""Weights + renderers, mirroring nerfstudio/cameras/rays.py:131-230 and nerfstudio/model_components/renderers.py.

The reference composes get_alpha -> get_weights_from_alphas -> RGBRenderer / DepthRenderer / SemanticRenderer /
AccumulationRenderer out of ~40 small PyTorch kernels; here the whole chain is one HIP kernel per direction
(one 64-lane wavefront per ray, DPP scans): ``neus_render``.  ``density_to_weights`` is RaySamples.get_weights.
The thin nn.Module wrappers keep the reference's class names / call signatures.
"""
from typing import Optional

import torch
from torch import nn

from sdfstudio_amd import _lib
from sdfstudio_amd.grad_slots import grad_target


class _DensityWeights(torch.autograd.Function):
    """RaySamples.get_weights (rays.py:146-167): w_i = (1 - exp(-d_i s_i)) exp(-sum_{j<i} d_j s_j)."""

    @staticmethod
    def forward(ctx, density, starts, ends):
        lib = _lib.load()
        density = density.contiguous()
        n, s = density.shape
        weights = torch.empty_like(density)
        _lib.check(lib.sdfhip_density_weights_forward(_lib.ptr(density), _lib.ptr(starts), _lib.ptr(ends), n, s,
                                                      _lib.ptr(weights), _lib.stream()), "density_weights_forward")
        ctx.save_for_backward(density, starts, ends)
        return weights

    @staticmethod
    def backward(ctx, wbar):
        density, starts, ends = ctx.saved_tensors
        lib = _lib.load()
        n, s = density.shape
        dbar = torch.empty_like(density)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_density_weights_backward(_lib.ptr(density), _lib.ptr(starts), _lib.ptr(ends), n, s,
                                                       kp(wbar), _lib.ptr(dbar), _lib.stream()),
                   "density_weights_backward")
        del kp
        return dbar, None, None


def density_to_weights(density: torch.Tensor, starts: torch.Tensor, ends: torch.Tensor) -> torch.Tensor:
    """density, starts, ends: [N,S] -> weights [N,S]."""
    return _DensityWeights.apply(density, starts.contiguous(), ends.contiguous())


def _variance_target(param, variance):
    """Where the render backward accumulates d L / d variance: the parameter's slot of the flat gradient buffer when there is one (zeroed
    by FlatGradients.zero; saves the fill and the hook's copy), else a fresh zero tensor."""
    if param is not None:
        return grad_target(param, zero_init=True)[0]
    return torch.zeros_like(variance)


class _NeusRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, grad, rgb, variance, dirs, starts, ends, background, cos_anneal):
        lib = _lib.load()
        n, s = starts.shape
        dev = starts.device
        sdf, grad, rgb = sdf.contiguous(), grad.contiguous(), rgb.contiguous()
        alpha = torch.empty(n, s, device=dev)
        weights = torch.empty(n, s, device=dev)
        out_rgb = torch.empty(n, 3, device=dev)
        depth_raw = torch.empty(n, device=dev)
        depth = torch.empty(n, device=dev)
        normal = torch.empty(n, 3, device=dev)
        acc = torch.empty(n, device=dev)
        minmax = torch.empty(2, device=dev)
        _lib.check(lib.sdfhip_neus_render_forward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends),
            _lib.ptr(variance), _lib.ptr(background), float(cos_anneal), n, s, _lib.ptr(alpha), _lib.ptr(weights),
            _lib.ptr(out_rgb), _lib.ptr(depth_raw), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(acc), _lib.ptr(minmax),
            _lib.stream()), "neus_render_forward")
        ctx.save_for_backward(sdf, grad, rgb, variance, dirs, starts, ends, alpha, weights, depth_raw, acc, minmax)
        ctx.background = background
        ctx.cos_anneal = float(cos_anneal)
        ctx.var_param = variance if (variance.is_leaf and variance.requires_grad) else None
        ctx.mark_non_differentiable(alpha)
        ctx.set_materialize_grads(False)  # unused heads (depth, normal, accumulation in training) arrive as None = NULL, not as zero fills
        return out_rgb, depth, normal, acc, weights, alpha

    @staticmethod
    def backward(ctx, rgb_bar, depth_bar, normal_bar, acc_bar, weights_bar, _alpha_bar):
        sdf, grad, rgb, variance, dirs, starts, ends, alpha, weights, depth_raw, acc, minmax = ctx.saved_tensors
        lib = _lib.load()
        n, s = starts.shape
        sdf_bar = torch.empty_like(sdf)
        grad_bar = torch.empty_like(grad)
        rgbs_bar = torch.empty_like(rgb)
        var_bar = _variance_target(ctx.var_param, variance)

        kp = _lib.Keep()  # cotangents may be stride-0 expands: their contiguous copies must all outlive the launch
        _lib.check(lib.sdfhip_neus_render_backward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends),
            _lib.ptr(variance), _lib.ptr(ctx.background), ctx.cos_anneal, n, s, _lib.ptr(alpha), _lib.ptr(weights),
            _lib.ptr(depth_raw), _lib.ptr(acc), _lib.ptr(minmax), kp(rgb_bar), kp(depth_bar),
            kp(normal_bar), kp(acc_bar), kp(weights_bar), _lib.ptr(sdf_bar), _lib.ptr(grad_bar),
            _lib.ptr(rgbs_bar), _lib.ptr(var_bar), _lib.stream()), "neus_render_backward")
        del kp
        return sdf_bar, grad_bar, rgbs_bar, var_bar, None, None, None, None, None


def neus_render(sdf, gradients, rgb, variance, directions, starts, ends, cos_anneal_ratio: float,
                background: Optional[torch.Tensor] = None):
    """Fused SDFField.get_alpha (sdf_field.py:476-525) + get_weights_from_alphas (rays.py:194-208) + renderers.

    sdf [N,S], gradients [N,S,3], rgb [N,S,3], variance [1] (deviation_network.variance), directions [N,3],
    starts/ends [N,S].  Returns rgb [N,3], depth [N] (expected, clipped as renderers.py:257), normal [N,3]
    (sum of w * normalize(grad)), accumulation [N], weights [N,S], alpha [N,S].
    """
    return _NeusRender.apply(sdf, gradients, rgb, variance, directions.contiguous(), starts.contiguous(),
                             ends.contiguous(), background, cos_anneal_ratio)


class _NeusRenderBg(torch.autograd.Function):
    """neus_render with NeuS-facto's background merge fused in (include/sdfhip.h: sdfhip_neus_render_bg_*): one launch each way where
    the per-head path (get_alpha -> merge -> get_weights_from_alphas -> four renderers, base_surface_model.py:266-310) costs ~50."""

    @staticmethod
    def forward(ctx, sdf, grad, rgb, variance, bg_density, bg_rgb, origins, dirs, starts, ends, background, cos_anneal):
        lib = _lib.load()
        n, s = starts.shape
        dev = starts.device
        sdf, grad, rgb, bg_density, bg_rgb = sdf.contiguous(), grad.contiguous(), rgb.contiguous(), bg_density.contiguous(), bg_rgb.contiguous()
        alpha, weights = torch.empty(n, s, device=dev), torch.empty(n, s, device=dev)
        out_rgb, normal = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        depth_raw, depth, acc = (torch.empty(n, device=dev) for _ in range(3))
        minmax = torch.empty(2, device=dev)
        rgb_merged = torch.empty(n, s, 3, device=dev)
        _lib.check(lib.sdfhip_neus_render_bg_forward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends), _lib.ptr(variance),
            _lib.ptr(background), float(cos_anneal), n, s, _lib.ptr(origins), _lib.ptr(bg_density), _lib.ptr(bg_rgb), _lib.ptr(alpha),
            _lib.ptr(weights), _lib.ptr(out_rgb), _lib.ptr(depth_raw), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(acc), _lib.ptr(minmax),
            _lib.ptr(rgb_merged), _lib.stream()), "neus_render_bg_forward")
        ctx.save_for_backward(sdf, grad, rgb, variance, bg_density, bg_rgb, origins, dirs, starts, ends, alpha, weights, depth_raw, acc, minmax)
        ctx.background = background
        ctx.cos_anneal = float(cos_anneal)
        ctx.var_param = variance if (variance.is_leaf and variance.requires_grad) else None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(alpha, rgb_merged)
        return out_rgb, depth, normal, acc, weights, alpha, rgb_merged

    @staticmethod
    def backward(ctx, rgb_bar, depth_bar, normal_bar, acc_bar, weights_bar, _alpha_bar, _merged_bar):
        sdf, grad, rgb, variance, bg_density, bg_rgb, origins, dirs, starts, ends, alpha, weights, depth_raw, acc, minmax = ctx.saved_tensors
        lib = _lib.load()
        n, s = starts.shape
        sdf_bar, grad_bar, rgbs_bar = torch.empty_like(sdf), torch.empty_like(grad), torch.empty_like(rgb)
        bgd_bar, bgc_bar = torch.empty_like(bg_density), torch.empty_like(bg_rgb)
        var_bar = _variance_target(ctx.var_param, variance)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_neus_render_bg_backward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(dirs), _lib.ptr(starts), _lib.ptr(ends), _lib.ptr(variance),
            _lib.ptr(ctx.background), ctx.cos_anneal, n, s, _lib.ptr(origins), _lib.ptr(bg_density), _lib.ptr(bg_rgb), _lib.ptr(alpha),
            _lib.ptr(weights), _lib.ptr(depth_raw), _lib.ptr(acc), _lib.ptr(minmax), kp(rgb_bar), kp(depth_bar), kp(normal_bar), kp(acc_bar),
            kp(weights_bar), _lib.ptr(sdf_bar), _lib.ptr(grad_bar), _lib.ptr(rgbs_bar), _lib.ptr(var_bar), _lib.ptr(bgd_bar), _lib.ptr(bgc_bar),
            _lib.stream()), "neus_render_bg_backward")
        del kp
        return sdf_bar, grad_bar, rgbs_bar, var_bar, bgd_bar, bgc_bar, None, None, None, None, None, None


def neus_render_bg(sdf, gradients, rgb, variance, bg_density, bg_rgb, origins, directions, starts, ends, cos_anneal_ratio: float,
                   background: Optional[torch.Tensor] = None):
    """neus_render + forward_background_field_and_merge (base_surface_model.py:266-290): samples that start outside the unit sphere take
    the background field's alpha (from bg_density [N,S]) and colour (bg_rgb [N,S,3]).  Returns neus_render's tuple - `alpha` is the
    merged alpha - plus the merged per-sample colour [N,S,3] (the reference's field_outputs[RGB]; carried for inspection, no gradient)."""
    return _NeusRenderBg.apply(sdf, gradients, rgb, variance, bg_density, bg_rgb, origins.contiguous(), directions.contiguous(),
                               starts.contiguous(), ends.contiguous(), background, cos_anneal_ratio)


class _VolsdfRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, grad, rgb, beta, starts, ends, background):
        lib = _lib.load()
        n, s = starts.shape
        dev = starts.device
        sdf, grad, rgb, beta = sdf.contiguous(), grad.contiguous(), rgb.contiguous(), beta.reshape(1).contiguous()
        density, weights = torch.empty(n, s, device=dev), torch.empty(n, s, device=dev)
        out_rgb, normal = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        depth_raw, depth, acc, bg_trans = (torch.empty(n, device=dev) for _ in range(4))
        minmax = torch.empty(2, device=dev)
        _lib.check(lib.sdfhip_volsdf_render_forward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(starts), _lib.ptr(ends), _lib.ptr(beta), _lib.ptr(background), n, s,
            _lib.ptr(density), _lib.ptr(weights), _lib.ptr(out_rgb), _lib.ptr(depth_raw), _lib.ptr(depth), _lib.ptr(normal), _lib.ptr(acc),
            _lib.ptr(bg_trans), _lib.ptr(minmax), _lib.stream()), "volsdf_render_forward")
        ctx.save_for_backward(sdf, grad, rgb, beta, starts, ends, density, weights, depth_raw, acc, bg_trans, minmax)
        ctx.background = background
        ctx.mark_non_differentiable(density)
        return out_rgb, depth, normal, acc, weights, density, bg_trans

    @staticmethod
    def backward(ctx, rgb_bar, depth_bar, normal_bar, acc_bar, weights_bar, _density_bar, bgt_bar):
        sdf, grad, rgb, beta, starts, ends, density, weights, depth_raw, acc, bg_trans, minmax = ctx.saved_tensors
        lib = _lib.load()
        n, s = starts.shape
        sdf_bar, grad_bar, rgbs_bar = torch.empty_like(sdf), torch.empty_like(grad), torch.empty_like(rgb)
        beta_bar = torch.zeros_like(beta)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_volsdf_render_backward(
            _lib.ptr(sdf), _lib.ptr(grad), _lib.ptr(rgb), _lib.ptr(starts), _lib.ptr(ends), _lib.ptr(beta), _lib.ptr(ctx.background), n, s,
            _lib.ptr(density), _lib.ptr(weights), _lib.ptr(depth_raw), _lib.ptr(acc), _lib.ptr(bg_trans), _lib.ptr(minmax), kp(rgb_bar),
            kp(depth_bar), kp(normal_bar), kp(acc_bar), kp(weights_bar), kp(bgt_bar), _lib.ptr(sdf_bar), _lib.ptr(grad_bar), _lib.ptr(rgbs_bar),
            _lib.ptr(beta_bar), _lib.stream()), "volsdf_render_backward")
        del kp
        return sdf_bar, grad_bar, rgbs_bar, beta_bar.view(ctx.saved_tensors[3].shape), None, None, None


def volsdf_render(sdf, gradients, rgb, beta, starts, ends, background: Optional[torch.Tensor] = None):
    """VolSDF's compositing in one launch (models/volsdf.py:62-79): LaplaceDensity (sdf_field.py:48-76) with beta [1] =
    laplace_density.get_beta(), get_weights (rays.py:146-167) and the four renderers.  sdf [N,S], gradients / rgb [N,S,3],
    starts / ends [N,S].  Returns rgb [N,3], depth [N] (expected, clipped as renderers.py:257), normal [N,3], accumulation [N],
    weights [N,S], density [N,S] (no gradient flows through this output) and the transmittance in front of the last sample [N]."""
    return _VolsdfRender.apply(sdf, gradients, rgb, beta, starts.contiguous(), ends.contiguous(), background)


def _sum_along_rays(values: torch.Tensor, ray_indices: Optional[torch.Tensor], num_rays: Optional[int]) -> torch.Tensor:
    """Sum over the samples of a ray: dim -2 of dense [..., S, D] samples, or - packed samples [P, D] with their ray index, as
    nerfacc.accumulate_along_rays (renderers.py:74-79,192-194) - an index_add into [num_rays, D]."""
    if ray_indices is not None and num_rays is not None:
        out = torch.zeros(int(num_rays), values.shape[-1], device=values.device, dtype=values.dtype)
        return out.index_add_(0, ray_indices.reshape(-1).long(), values.reshape(-1, values.shape[-1]))
    return torch.sum(values, dim=-2)


class RGBRenderer(nn.Module):
    """renderers.py:42-118: sum_s w rgb + bg (1 - sum_s w); clamp to [0,1] in eval.  background_color: an RGB tensor, "random" (the
    reference's default: a fresh uniform colour per ray and call, :86-87), "last_sample" (the colour of the ray's last sample, :84-85;
    dense samples only), or None (black: an extension - the reference asserts a tensor)."""

    def __init__(self, background_color="random") -> None:
        super().__init__()
        self.background_color = background_color

    @classmethod
    def combine_rgb(cls, rgb: torch.Tensor, weights: torch.Tensor, background_color="random", ray_indices: Optional[torch.Tensor] = None,
                    num_rays: Optional[int] = None) -> torch.Tensor:
        """renderers.py:53-92 (dense [..., S, .] samples, or packed samples with ray_indices / num_rays)."""
        packed = ray_indices is not None and num_rays is not None
        if packed and isinstance(background_color, str) and background_color == "last_sample":
            raise NotImplementedError("Background color 'last_sample' not implemented for packed samples.")
        comp = _sum_along_rays(weights * rgb, ray_indices, num_rays)
        acc = _sum_along_rays(weights, ray_indices, num_rays)
        bg = background_color
        if isinstance(bg, str):
            if bg == "last_sample":
                bg = rgb[..., -1, :]
            elif bg == "random":
                bg = torch.rand_like(comp)
            else:
                raise ValueError(f"background_color must be an RGB tensor, 'random' or 'last_sample', not {bg!r}")
        if isinstance(bg, torch.Tensor):
            comp = comp + bg.to(comp) * (1.0 - acc)
        return comp

    def forward(self, rgb: torch.Tensor, weights: torch.Tensor, ray_indices: Optional[torch.Tensor] = None,
                num_rays: Optional[int] = None) -> torch.Tensor:
        comp = self.combine_rgb(rgb, weights, background_color=self.background_color, ray_indices=ray_indices, num_rays=num_rays)
        if not self.training:
            comp = comp.clamp(0.0, 1.0)
        return comp


class AccumulationRenderer(nn.Module):
    """renderers.py:171-197."""

    def forward(self, weights: torch.Tensor, ray_indices: Optional[torch.Tensor] = None, num_rays: Optional[int] = None) -> torch.Tensor:
        return _sum_along_rays(weights, ray_indices, num_rays)


class DepthRenderer(nn.Module):
    """renderers.py:200-261: method 'expected' (what the surface models use; fused into the compositing kernels on the training path) and
    'median' (:234-244: the mid point of the sample at which the cumulative weight reaches 0.5; not differentiable, per-head mirror only)."""

    def __init__(self, method: str = "median") -> None:  # the reference's default (renderers.py:211); the surface models pass "expected"
        super().__init__()
        if method not in ("expected", "median"):
            raise NotImplementedError(f"depth method {method!r} (the reference has 'expected' and 'median')")
        self.method = method

    def forward(self, weights: torch.Tensor, ray_samples, ray_indices=None, num_rays=None) -> torch.Tensor:
        steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
        if self.method == "median":
            if ray_indices is not None and num_rays is not None:
                raise NotImplementedError("Median depth calculation is not implemented for packed samples.")  # the reference's message
            cumulative = torch.cumsum(weights[..., 0], dim=-1)
            split = torch.full((*weights.shape[:-2], 1), 0.5, device=weights.device, dtype=cumulative.dtype)
            index = torch.searchsorted(cumulative, split, side="left").clamp_(0, steps.shape[-2] - 1)
            return torch.gather(steps[..., 0], dim=-1, index=index)
        if ray_indices is not None and num_rays is not None:
            # renderers.py:249-253, packed samples (nerfacc.accumulate_along_rays twice): weights / steps are [P, 1]; a per-head mirror
            # (the NeuS-acc training path composites all heads in ONE segmented kernel: models/neus_acc.py, accumulate_along_rays)
            w, st = weights.reshape(-1, 1), steps.reshape(-1, 1)
            depth = torch.zeros(int(num_rays), 1, device=w.device, dtype=w.dtype).index_add_(0, ray_indices.reshape(-1).long(), w * st)
            acc = torch.zeros(int(num_rays), 1, device=w.device, dtype=w.dtype).index_add_(0, ray_indices.reshape(-1).long(), w)
            depth = depth / (acc + 1e-10)
        else:
            depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + 1e-10)
        return torch.clip(depth, steps.min(), steps.max())


class SemanticRenderer(nn.Module):
    """renderers.py:284-295 (used for normals by SurfaceModel)."""

    def forward(self, semantics: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
        return torch.sum(weights * semantics, dim=-2)


# ---- packed samples (NeuS-acc): the two nerfacc compositing operators the reference calls (models/neus_acc.py:103-121)
class _PackedWeights(torch.autograd.Function):
    """nerfacc.render_weight_from_alpha: w_i = alpha_i prod_{j<i, same ray} (1 - alpha_j) on packed samples."""

    @staticmethod
    def forward(ctx, alpha, offsets, counts):
        lib = _lib.load()
        alpha = alpha.contiguous()
        # zeros, not empty: the kernel writes the rays' segments only; with bounded packed arrays (ray_samplers.march_occupancy_grid(capacity))
        # the tail behind the last segment is a filler whose weight - and, in backward, whose alpha cotangent - must be exactly zero
        weights, trans = torch.zeros_like(alpha), torch.empty_like(alpha)
        _lib.check(lib.sdfhip_packed_weights_forward(_lib.ptr(alpha), offsets.data_ptr(), counts.data_ptr(), counts.shape[0],
                                                     _lib.ptr(weights), _lib.ptr(trans), _lib.stream()), "packed_weights_forward")
        ctx.save_for_backward(alpha, weights, trans, offsets, counts)
        return weights

    @staticmethod
    def backward(ctx, wbar):
        alpha, weights, trans, offsets, counts = ctx.saved_tensors
        lib = _lib.load()
        abar = torch.zeros_like(alpha)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_packed_weights_backward(_lib.ptr(alpha), _lib.ptr(weights), _lib.ptr(trans), kp(wbar), offsets.data_ptr(),
                                                      counts.data_ptr(), counts.shape[0], _lib.ptr(abar), _lib.stream()),
                   "packed_weights_backward")
        del kp
        return abar, None, None


class _PackedAccumulate(torch.autograd.Function):
    """nerfacc.accumulate_along_rays: out[r] = sum over ray r's samples of w_i values_i (deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, weights, values, offsets, counts, ray_indices):
        lib = _lib.load()
        weights = weights.contiguous()
        values = None if values is None else values.contiguous()
        dim = 1 if values is None else values.shape[-1]
        out = torch.empty(counts.shape[0], dim, device=weights.device)
        _lib.check(lib.sdfhip_packed_accumulate(_lib.ptr(weights), _lib.ptr(values), offsets.data_ptr(), counts.data_ptr(), counts.shape[0],
                                                dim, _lib.ptr(out), _lib.stream()), "packed_accumulate")
        ctx.save_for_backward(weights, values, ray_indices)
        ctx.has_values = values is not None
        return out

    @staticmethod
    def backward(ctx, obar):
        weights, values, ray_indices = ctx.saved_tensors
        g = obar[ray_indices]  # [P, D]
        if not ctx.has_values:
            return g[:, 0], None, None, None, None
        return (g * values).sum(-1), weights[:, None] * g, None, None, None


def render_weight_from_alpha(alphas: torch.Tensor, packed_info: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """alphas [P,1] (or [P]) -> weights of the same shape; packed_info [N,2] = (offset, count) int64, counts [N] int32."""
    w = _PackedWeights.apply(alphas.reshape(-1), packed_info[:, 0].contiguous(), counts)
    return w.view(alphas.shape)


def accumulate_along_rays(weights: torch.Tensor, ray_indices: torch.Tensor, values: Optional[torch.Tensor], packed_info: torch.Tensor,
                          counts: torch.Tensor) -> torch.Tensor:
    """weights [P,1] (or [P]), values [P,D] or None -> [N,D] ([N,1])."""
    v = None if values is None else values.reshape(values.shape[0], -1)
    return _PackedAccumulate.apply(weights.reshape(-1), v, packed_info[:, 0].contiguous(), counts, ray_indices)
