"""Ray samplers mirroring nerfstudio/model_components/ray_samplers.py (Sampler :32, SpacedSampler :55,
UniformLinDispPiecewiseSampler :221, PDFSampler :250, ProposalNetworkSampler :497).

Each sampler is one HIP kernel launch (the reference issues 10-25 small PyTorch kernels each).  Sampler outputs
carry no gradient (bins.detach(), ray_samplers.py:358).  Stratified jitter uses a single draw per ray
(``single_jitter=True``, the neus-facto setting, neus_facto.py:145); tests inject the draw to compare with the oracle.
"""
from typing import Callable, List, Optional, Tuple

import ctypes

import torch
from torch import nn

from sdfstudio_amd import _lib
from sdfstudio_amd.cameras.rays import RayBundle, RaySamples


_SPACINGS = {
    # name: (kernel id (include/sdfhip.h SDFHIP_SPACING_*), spacing_fn, spacing_fn_inv)  -- ray_samplers.py:130-247
    "piecewise": (0, lambda v: torch.where(v < 1, v / 2, 1 - 1 / (2 * v)), lambda v: torch.where(v < 0.5, 2 * v, 1 / (2 - 2 * v))),
    "uniform": (1, lambda v: v, lambda v: v),
    "lindisp": (2, lambda v: 1 / v, lambda v: 1 / v),
    "sqrt": (3, torch.sqrt, lambda v: v ** 2),
    "log": (4, torch.log, torch.exp),
}


def _spacing_to_euclidean(kind: str, nears: torch.Tensor, fars: torch.Tensor) -> Callable:
    """spacing_to_euclidean_fn of a SpacedSampler (ray_samplers.py:115-117): x -> fn_inv(x s_far + (1 - x) s_near)."""
    _, fn, inv = _SPACINGS[kind]
    if kind == "uniform":
        return lambda x: x * fars + (1 - x) * nears

    def to_euclidean(x):
        return inv(x * fn(fars) + (1 - x) * fn(nears))

    return to_euclidean


def _piecewise_to_euclidean(nears: torch.Tensor, fars: torch.Tensor) -> Callable:
    return _spacing_to_euclidean("piecewise", nears, fars)


class Sampler(nn.Module):
    """ray_samplers.py:32-52."""

    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)


def _make_samples(ray_bundle: RayBundle, bins, starts, ends, kind: str = "piecewise") -> RaySamples:
    return ray_bundle.get_ray_samples(
        bin_starts=starts[..., None], bin_ends=ends[..., None], spacing_starts=bins[:, :-1, None],
        spacing_ends=bins[:, 1:, None], spacing_to_euclidean_fn=_spacing_to_euclidean(kind, ray_bundle.nears, ray_bundle.fars),
        flat_bins=bins,
    )


def _make_uniform_samples(ray_bundle: RayBundle, bins, starts, ends) -> RaySamples:
    return _make_samples(ray_bundle, bins, starts, ends, "uniform")


def _identify_spacing(spacing_fn: Callable, spacing_fn_inv: Callable) -> str:
    """Which built spacing a (spacing_fn, spacing_fn_inv) pair of the reference's SpacedSampler constructor (ray_samplers.py:66-78) is: the
    kernel holds the five pairs the reference's own samplers use (:130-247), recognised by their values at a few points."""
    t = torch.tensor([0.25, 0.75, 1.0, 2.0, 7.5], dtype=torch.float64)
    for name, (_, fn, inv) in _SPACINGS.items():
        try:
            if torch.allclose(torch.as_tensor(spacing_fn(t), dtype=torch.float64), fn(t), rtol=1e-12, atol=0.0) and \
                    torch.allclose(torch.as_tensor(spacing_fn_inv(fn(t)), dtype=torch.float64), inv(fn(t)), rtol=1e-12, atol=0.0):
                return name
        except Exception:  # noqa: BLE001 - a callable that does not take tensors is simply not one of the five
            continue
    raise NotImplementedError("SpacedSampler: the native sampler is built for the reference's own spacing functions (uniform, 1 / x, sqrt, "
                              "log, uniform + linear-disparity piecewise: ray_samplers.py:130-247), not for arbitrary callables")


class SpacedSampler(Sampler):
    """ray_samplers.py:55-127: stratified bins in a spacing domain mapped back to euclidean distances, ONE kernel launch
    (sdfhip_sample_spacing) for the whole family.  Constructor as the reference's (spacing_fn, spacing_fn_inv, num_samples,
    train_stratified, single_jitter); `spacing` names the pair directly.  The reference's default is per-bin-edge jitter
    (single_jitter=False, ray_samplers.py:105-113), neus-facto uses one draw per ray."""

    spacing = "uniform"

    def __init__(self, spacing_fn: Optional[Callable] = None, spacing_fn_inv: Optional[Callable] = None, num_samples: Optional[int] = None,
                 train_stratified=True, single_jitter=False, spacing: Optional[str] = None) -> None:
        super().__init__(num_samples=num_samples)
        if spacing is not None:
            self.spacing = spacing
        elif spacing_fn is not None or spacing_fn_inv is not None:
            self.spacing = _identify_spacing(spacing_fn, spacing_fn_inv)
        if self.spacing not in _SPACINGS:
            raise ValueError(f"unknown spacing {self.spacing!r}; built: {sorted(_SPACINGS)}")
        self.spacing_fn, self.spacing_fn_inv = _SPACINGS[self.spacing][1], _SPACINGS[self.spacing][2]
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        self.jitter_override: Optional[torch.Tensor] = None  # tests inject the draw: [N] / [N,1] (single) or [N,S+1] (per edge)

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None) -> RaySamples:
        lib = _lib.load()
        assert ray_bundle is not None and ray_bundle.nears is not None and ray_bundle.fars is not None
        s = num_samples or self.num_samples
        assert s is not None
        n = len(ray_bundle)
        dev = ray_bundle.origins.device
        kp = _lib.Keep()
        jitter = None
        if self.train_stratified and self.training:
            shape = (n,) if self.single_jitter else (n, s + 1)
            jitter = self.jitter_override if self.jitter_override is not None else torch.rand(shape, device=dev)
            jitter = jitter.reshape(shape)
        bins = torch.empty(n, s + 1, device=dev)
        starts = torch.empty(n, s, device=dev)
        ends = torch.empty(n, s, device=dev)
        _lib.check(lib.sdfhip_sample_spacing(_SPACINGS[self.spacing][0], kp(ray_bundle.nears.reshape(-1)), kp(ray_bundle.fars.reshape(-1)),
                                             kp(jitter), 0 if self.single_jitter else 1, n, s, _lib.ptr(bins), _lib.ptr(starts),
                                             _lib.ptr(ends), _lib.stream()), "sample_spacing")
        del kp
        return _make_samples(ray_bundle, bins, starts, ends, self.spacing)


class UniformSampler(SpacedSampler):
    """ray_samplers.py:130-151."""

    spacing = "uniform"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter, spacing=self.spacing)


class LinearDisparitySampler(SpacedSampler):
    """ray_samplers.py:154-175 (the background sampler of the surface models, base_surface_model.py:214)."""

    spacing = "lindisp"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter, spacing=self.spacing)


class SqrtSampler(SpacedSampler):
    """ray_samplers.py:178-198."""

    spacing = "sqrt"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter, spacing=self.spacing)


class LogSampler(SpacedSampler):
    """ray_samplers.py:201-218."""

    spacing = "log"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        super().__init__(num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter, spacing=self.spacing)


class UniformLinDispPiecewiseSampler(SpacedSampler):
    """ray_samplers.py:221-247: first half uniform, second half linear in disparity."""

    spacing = "piecewise"

    def __init__(self, num_samples: Optional[int] = None, train_stratified=True, single_jitter=False) -> None:
        # defaults as the reference's (ray_samplers.py:235-240); every caller on the path passes single_jitter (neus_facto.py:145)
        super().__init__(num_samples=num_samples, train_stratified=train_stratified, single_jitter=single_jitter, spacing=self.spacing)


class NeuSSampler(Sampler):
    """ray_samplers.py:815-944.  Each up-sampling step (merge of the new sdf values, alpha with a fixed inverse variance,
    weights, PDF resampling, sorted merge) is ONE kernel launch (sdfhip_neus_upsample) plus the field's no-grad sdf
    evaluation at the new samples; the reference issues ~60 small PyTorch kernels per step."""

    def __init__(self, num_samples: int = 64, num_samples_importance: int = 64, num_samples_outside: int = 32,
                 num_upsample_steps: int = 4, base_variance: float = 64, single_jitter: bool = True) -> None:
        super().__init__()
        self.single_jitter = single_jitter
        self.num_samples = num_samples
        self.num_samples_importance = num_samples_importance
        self.num_samples_outside = num_samples_outside  # unused by the reference as well (ray_samplers.py:888-893)
        self.num_upsample_steps = num_upsample_steps
        self.base_variance = base_variance
        self.uniform_sampler = UniformSampler(single_jitter=single_jitter)
        self.jitter_overrides: Optional[List[torch.Tensor]] = None  # tests: one draw per up-sampling step ([N], or [N, n_new + 1])

    def upsample_step(self, ray_bundle: RayBundle, bins, sdf_a, sdf_b, index, n_new: int, inv_s: float, jitter):
        """One reference loop iteration; returns (sdf_merged, new_bins, new_starts, new_ends, merged_bins, merged_index,
        merged_starts, merged_ends)."""
        lib = _lib.load()
        n, s1 = bins.shape
        s = s1 - 1
        dev = bins.device
        s_a = sdf_a.shape[1]
        s_b = 0 if sdf_b is None else sdf_b.shape[1]
        assert s_a + s_b == s
        nears = ray_bundle.nears.reshape(-1).contiguous()
        fars = ray_bundle.fars.reshape(-1).contiguous()
        sdf_m = torch.empty(n, s, device=dev)
        new_bins = torch.empty(n, n_new + 1, device=dev)
        new_starts = torch.empty(n, n_new, device=dev)
        new_ends = torch.empty(n, n_new, device=dev)
        m_bins = torch.empty(n, s + n_new + 1, device=dev)
        m_index = torch.empty(n, s + n_new, device=dev, dtype=torch.int32)
        m_starts = torch.empty(n, s + n_new, device=dev)
        m_ends = torch.empty(n, s + n_new, device=dev)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_neus_upsample(
            kp(bins), kp(sdf_a), kp(sdf_b),
            None if index is None else index.data_ptr(), _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(jitter),
            0 if jitter is None or jitter.dim() == 1 else 1, n, s_a, s_b, n_new,
            float(inv_s), _lib.ptr(sdf_m), _lib.ptr(new_bins), _lib.ptr(new_starts), _lib.ptr(new_ends), _lib.ptr(m_bins),
            m_index.data_ptr(), _lib.ptr(m_starts), _lib.ptr(m_ends), _lib.stream()), "neus_upsample")
        del kp
        return sdf_m, new_bins, new_starts, new_ends, m_bins, m_index, m_starts, m_ends

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, sdf_fn: Optional[Callable] = None,
                             ray_samples: Optional[RaySamples] = None) -> RaySamples:
        assert ray_bundle is not None and sdf_fn is not None
        if ray_samples is None:
            ray_samples = self.uniform_sampler(ray_bundle, num_samples=self.num_samples)
        n = len(ray_bundle)
        dev = ray_bundle.origins.device
        bins = ray_samples.flat_bins
        n_new = self.num_samples_importance // self.num_upsample_steps
        new_samples = ray_samples
        sdf, index = None, None
        m_starts, m_ends = ray_samples.flat_starts, ray_samples.flat_ends
        for it in range(self.num_upsample_steps):
            with torch.no_grad():
                new_sdf = sdf_fn(new_samples)[..., 0]
            jitter = None
            if self.training:
                shape = (n,) if self.single_jitter else (n, n_new + 1)  # PDFSampler's draw, ray_samplers.py:321-330
                jitter = (self.jitter_overrides[it] if self.jitter_overrides is not None else torch.rand(shape, device=dev))
                jitter = jitter.reshape(shape).contiguous()
            if sdf is None:
                sdf_a, sdf_b = new_sdf, None
            else:
                sdf_a, sdf_b = sdf, new_sdf
            sdf, new_bins, new_starts, new_ends, bins, index, m_starts, m_ends = self.upsample_step(
                ray_bundle, bins, sdf_a, sdf_b, index, n_new, self.base_variance * 2 ** it, jitter)
            new_samples = _make_uniform_samples(ray_bundle, new_bins, new_starts, new_ends)
        return _make_uniform_samples(ray_bundle, bins, m_starts, m_ends)


class ErrorBoundedSampler(Sampler):
    """ray_samplers.py:581-788 (VolSDF Algorithm 1).  Per outer iteration: the field's no-grad sdf at the new samples, ONE kernel
    for merge + d* + the beta bisection (11 error-bound evaluations) + weights (sdfhip_volsdf_bound_step), then one PDF kernel and
    one merge kernel; the reference issues several hundred small PyTorch kernels per iteration.  The loop-exit decision
    (`beta.max() > beta0`, :671) is a host read of one int, as in the reference."""

    def __init__(self, num_samples: int = 64, num_samples_eval: int = 128, num_samples_extra: int = 32, eps: float = 0.1,
                 beta_iters: int = 10, max_total_iters: int = 5, add_tiny: float = 1e-6, single_jitter: bool = False) -> None:
        super().__init__()
        self.num_samples, self.num_samples_eval, self.num_samples_extra = num_samples, num_samples_eval, num_samples_extra
        self.eps, self.beta_iters, self.max_total_iters, self.add_tiny = eps, beta_iters, max_total_iters, add_tiny
        self.single_jitter = single_jitter
        self.uniform_sampler = UniformSampler(single_jitter=single_jitter)
        self.jitter_queue: Optional[List[torch.Tensor]] = None  # tests: the torch.rand draws in call order

    def _draw(self, n, m, dev):
        if not self.training:
            return None
        shape = (n,) if self.single_jitter else (n, m + 1)
        if self.jitter_queue is not None:
            return self.jitter_queue.pop(0).to(dev).reshape(shape).contiguous()
        return torch.rand(shape, device=dev)

    def _pdf(self, ray_bundle, weights, bins, s_out):
        lib = _lib.load()
        n, s_in = weights.shape
        dev = weights.device
        nears, fars = ray_bundle.nears.reshape(-1).contiguous(), ray_bundle.fars.reshape(-1).contiguous()
        jitter = self._draw(n, s_out, dev)
        out_bins = torch.empty(n, s_out + 1, device=dev)
        starts = torch.empty(n, s_out, device=dev)
        ends = torch.empty(n, s_out, device=dev)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_sample_pdf_uniform(kp(weights), kp(bins), _lib.ptr(nears),
                                                 _lib.ptr(fars), _lib.ptr(jitter), 0 if self.single_jitter else 1, n, s_in, s_out,
                                                 1e-5, _lib.ptr(out_bins), _lib.ptr(starts), _lib.ptr(ends), _lib.stream()),
                   "sample_pdf_uniform")
        del kp
        return out_bins, starts, ends

    def merge(self, ray_bundle, bins_1, bins_2):
        """merge_ray_samples (:757-786) on spacing bins -> (bins, index, starts, ends)."""
        lib = _lib.load()
        n, s1, s2 = bins_1.shape[0], bins_1.shape[1] - 1, bins_2.shape[1] - 1
        dev = bins_1.device
        nears, fars = ray_bundle.nears.reshape(-1).contiguous(), ray_bundle.fars.reshape(-1).contiguous()
        m_bins = torch.empty(n, s1 + s2 + 1, device=dev)
        m_index = torch.empty(n, s1 + s2, device=dev, dtype=torch.int32)
        m_starts = torch.empty(n, s1 + s2, device=dev)
        m_ends = torch.empty(n, s1 + s2, device=dev)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_merge_uniform(kp(bins_1), kp(bins_2), _lib.ptr(nears), _lib.ptr(fars),
                                            n, s1, s2, _lib.ptr(m_bins), m_index.data_ptr(), _lib.ptr(m_starts), _lib.ptr(m_ends),
                                            _lib.stream()), "merge_uniform")
        del kp
        return m_bins, m_index, m_starts, m_ends

    def bound_step(self, ray_bundle, bins, sdf_a, sdf_b, index, beta, beta0):
        """One Algorithm-1 iteration up to the weights: (sdf_merged, beta_out, weights, err_weights, not_converged)."""
        lib = _lib.load()
        n, s = bins.shape[0], bins.shape[1] - 1
        dev = bins.device
        s_a = sdf_a.shape[1]
        s_b = 0 if sdf_b is None else sdf_b.shape[1]
        assert s_a + s_b == s
        nears, fars = ray_bundle.nears.reshape(-1).contiguous(), ray_bundle.fars.reshape(-1).contiguous()
        sdf_m = torch.empty(n, s, device=dev)
        beta_out = torch.empty(n, device=dev)
        weights = torch.empty(n, s, device=dev)
        err_w = torch.empty(n, s, device=dev)
        flag = torch.zeros(1, device=dev, dtype=torch.int32)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_volsdf_bound_step(
            kp(bins), kp(sdf_a), kp(sdf_b),
            None if index is None else index.data_ptr(), _lib.ptr(nears), _lib.ptr(fars), kp(beta),
            kp(beta0.reshape(1)), n, s_a, s_b, float(self.eps), int(self.beta_iters), _lib.ptr(sdf_m),
            _lib.ptr(beta_out), _lib.ptr(weights), _lib.ptr(err_w), flag.data_ptr(), _lib.stream()), "volsdf_bound_step")
        del kp
        return sdf_m, beta_out, weights, err_w, flag

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fn: Optional[Callable] = None,
                             sdf_fn: Optional[Callable] = None, return_eikonal_points: bool = True):
        assert ray_bundle is not None and density_fn is not None and sdf_fn is not None
        import math

        n = len(ray_bundle)
        dev = ray_bundle.origins.device
        beta0 = density_fn.get_beta().detach().float()
        self.uniform_sampler.train(self.training)
        self.uniform_sampler.jitter_override = self._draw(n, self.num_samples_eval, dev)
        ray_samples = self.uniform_sampler(ray_bundle, num_samples=self.num_samples_eval)
        bins = ray_samples.flat_bins
        deltas = ray_samples.flat_ends - ray_samples.flat_starts
        beta = torch.sqrt((1.0 / (4.0 * math.log(self.eps + 1.0))) * (deltas ** 2.0).sum(-1))  # Lemma 2 (:625-626)
        total, not_converge = 0, True
        sdf, index, new_samples = None, None, ray_samples
        while not_converge and total < self.max_total_iters:
            with torch.no_grad():
                new_sdf = sdf_fn(new_samples)[..., 0]
            sdf_a, sdf_b = (new_sdf, None) if sdf is None else (sdf, new_sdf)
            sdf, beta, weights, err_w, flag = self.bound_step(ray_bundle, bins, sdf_a, sdf_b, index, beta, beta0)
            total += 1
            not_converge = bool(flag.item())  # `beta.max() > beta0` (:671): the reference's host decision
            if not_converge and total < self.max_total_iters:
                new_bins, new_starts, new_ends = self._pdf(ray_bundle, err_w, bins, self.num_samples_eval)
                new_samples = _make_uniform_samples(ray_bundle, new_bins, new_starts, new_ends)
                bins, index, _, _ = self.merge(ray_bundle, bins, new_bins)
            else:
                bins, f_starts, f_ends = self._pdf(ray_bundle, weights, bins, self.num_samples)
        self.last_total_iters = total  # outer iterations of Algorithm 1 taken = device -> host decisions of this call (bench.py reports it)
        out = _make_uniform_samples(ray_bundle, bins, f_starts, f_ends)
        points = None
        if return_eikonal_points:
            # :685-689: random near-surface points (unused by the models: base_surface_model.py:343-345)
            mid = out.frustums.get_positions().reshape(-1, 3)
            points = mid[torch.randint(mid.shape[0], (n * 10,), device=dev)]
        if self.num_samples_extra > 0:
            self.uniform_sampler.jitter_override = self._draw(n, self.num_samples_extra, dev)
            extra = self.uniform_sampler(ray_bundle, num_samples=self.num_samples_extra)
            bins, _, m_starts, m_ends = self.merge(ray_bundle, bins, extra.flat_bins)
            out = _make_uniform_samples(ray_bundle, bins, m_starts, m_ends)
        return (out, points) if return_eikonal_points else out


class PDFSampler(Sampler):
    """ray_samplers.py:250-370.  The inverse-CDF resampling is one kernel (sdfhip_sample_pdf_spacing) in the spacing domain of the
    incoming samples, for both jitter modes; with include_original=True (the reference's default, though every caller on the
    SDF path passes False: :525, :835, :601) the new bins are merged with the existing ones by a sort, as the reference does (:355)."""

    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01, spacing: str = "piecewise") -> None:
        super().__init__(num_samples=num_samples)
        if spacing not in _SPACINGS:
            raise ValueError(f"unknown spacing {spacing!r}; built: {sorted(_SPACINGS)}")
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        self.include_original = include_original
        self.histogram_padding = histogram_padding
        self.spacing = spacing  # the spacing domain the incoming ray samples' bins live in (RaySamples carry only the closure)
        self.jitter_override: Optional[torch.Tensor] = None

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: Optional[torch.Tensor] = None, num_samples: Optional[int] = None, eps: float = 1e-5,
                             anneal: float = 1.0) -> RaySamples:
        lib = _lib.load()
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        assert weights is not None, "weights must be provided"
        if eps != 1e-5:  # ray_samplers.py:281,307: the padding threshold of the weight sum is a constant of the kernel
            raise NotImplementedError("PDFSampler: eps is fixed at the reference's default 1e-5 in pdf_sample_kernel")
        s_out = num_samples or self.num_samples
        w = weights[..., 0] if weights.dim() == 3 else weights
        w = w.detach()
        n, s_in = w.shape
        dev = w.device
        bins_in = ray_samples.flat_bins
        if bins_in is None:
            bins_in = torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], dim=-1)
        jitter = None
        if self.train_stratified and self.training:
            shape = (n,) if self.single_jitter else (n, s_out + 1)
            jitter = self.jitter_override if self.jitter_override is not None else torch.rand(shape, device=dev)
            jitter = jitter.reshape(shape)
        bins = torch.empty(n, s_out + 1, device=dev)
        starts = torch.empty(n, s_out, device=dev)
        ends = torch.empty(n, s_out, device=dev)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_sample_pdf_spacing(_SPACINGS[self.spacing][0], kp(w), kp(bins_in), kp(ray_bundle.nears.reshape(-1)),
                                                 kp(ray_bundle.fars.reshape(-1)), kp(jitter), 0 if self.single_jitter else 1, n, s_in,
                                                 s_out, float(anneal), float(self.histogram_padding), _lib.ptr(bins), _lib.ptr(starts),
                                                 _lib.ptr(ends), _lib.stream()), "sample_pdf_spacing")
        del kp
        if self.include_original:
            bins, _ = torch.sort(torch.cat([bins_in, bins], -1), -1)  # ray_samplers.py:354-355
            eu = _spacing_to_euclidean(self.spacing, ray_bundle.nears, ray_bundle.fars)(bins)
            starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
        return _make_samples(ray_bundle, bins, starts, ends, self.spacing)


class ProposalNetworkSampler(Sampler):
    """ray_samplers.py:497-578.  density_fns take a RaySamples (fused midpoint + contraction + grid + MLP kernel)."""

    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, use_uniform_sampler: bool = False, single_jitter: bool = False,
                 update_sched: Callable = lambda x: 1) -> None:
        super().__init__()
        if num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        # ray_samplers.py:517-522; the PDF kernel resamples in the spacing domain of the initial sampler's bins
        if use_uniform_sampler:
            self.initial_sampler = UniformSampler(single_jitter=single_jitter)
        else:
            self.initial_sampler = UniformLinDispPiecewiseSampler(single_jitter=single_jitter)
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter,
                                      spacing="uniform" if use_uniform_sampler else "piecewise")
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fns: Optional[List[Callable]] = None):
        assert ray_bundle is not None
        assert density_fns is not None
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples)
            else:
                # torch.pow(weights, anneal) (ray_samplers.py:562) is folded into the PDF kernel
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=num_samples,
                                               anneal=self._anneal)
            if is_prop:
                if updated:
                    density = density_fns[i_level](ray_samples)
                else:
                    with torch.no_grad():
                        density = density_fns[i_level](ray_samples)
                weights = ray_samples.get_weights(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        return ray_samples, weights_list, ray_samples_list


class UniSurfSampler(Sampler):
    """ray_samplers.py:947-1138: UniSurf's surface-guided sampler - the reference's ray / surface root finder ("sphere tracing" in
    the north star's words).  Per call: num_marching_steps uniform samples (one kernel) -> sdf at them (the field's no-grad sdf
    kernel) -> occupancy weights -> PDF importance samples (kernel) + outside samples (kernel) merged (kernel) -> first
    outside-to-inside sign change, interpolated depth and shrunk interval per ray (sdfhip_surface_root) -> uniform samples in that
    interval (kernel) -> merge in euclidean space (kernel).  The occupancy -> weights step is the reference's torch cumprod."""

    def __init__(self, num_samples_interval: int = 64, num_samples_outside: int = 32, num_samples_importance: int = 32,
                 num_marching_steps: int = 256, num_secant_steps: int = 8, interval_start: float = 0.25, interval_end: float = 0.0125,
                 interval_decay: float = 0.00005, single_jitter: bool = False) -> None:
        super().__init__()
        self.num_samples_interval, self.num_samples_outside = num_samples_interval, num_samples_outside
        self.num_samples_importance, self.num_marching_steps = num_samples_importance, num_marching_steps
        self.num_secant_steps = num_secant_steps  # unused by the reference as well (secant_method raises, :1132-1138)
        self.interval_start, self.interval_end, self.interval_decay = interval_start, interval_end, interval_decay
        self.single_jitter = single_jitter
        self.uniform_sampler = UniformSampler(single_jitter=single_jitter)
        self.outside_sampler = UniformSampler(single_jitter=single_jitter)
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter, histogram_padding=1e-5, spacing="uniform")
        self.error_bounded_sampler = ErrorBoundedSampler()  # for its merge
        self._step = 0
        self.delta = self.interval_start
        self.jitter_overrides: Optional[List[torch.Tensor]] = None  # tests: the four draws in call order

    def step_cb(self, step):
        """:987-990."""
        import math

        self._step = step
        self.delta = max(self.interval_start * math.exp(-1 * self.interval_decay * self._step), self.interval_end)

    def find_surface(self, ray_bundle: RayBundle, ray_samples: RaySamples, sdf: torch.Tensor):
        """(mask [N] bool, z [N], new nears [N], new fars [N]) - :1037-1075."""
        lib = _lib.load()
        n, s = ray_samples.flat_starts.shape
        dev = sdf.device
        mask = torch.empty(n, dtype=torch.int32, device=dev)
        z, nn, nf = torch.empty(n, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
        kp = _lib.Keep()
        _lib.check(lib.sdfhip_surface_root(kp(sdf.reshape(n, s)), kp(ray_samples.flat_starts), kp(ray_bundle.nears.reshape(-1)),
                                           kp(ray_bundle.fars.reshape(-1)), n, s, float(self.delta), mask.data_ptr(), _lib.ptr(z),
                                           _lib.ptr(nn), _lib.ptr(nf), _lib.stream()), "surface_root")
        del kp
        return mask.bool(), z, nn, nf

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, occupancy_fn: Optional[Callable] = None,
                             sdf_fn: Optional[Callable] = None, return_surface_points: bool = False):
        assert ray_bundle is not None and sdf_fn is not None and occupancy_fn is not None
        draws = list(self.jitter_overrides) if self.jitter_overrides is not None else [None] * 4
        self.uniform_sampler.jitter_override = draws[0]
        ray_samples = self.uniform_sampler(ray_bundle, num_samples=self.num_marching_steps)
        with torch.no_grad():
            sdf = sdf_fn(ray_samples)
        weights = ray_samples.get_weights_from_alphas(occupancy_fn(sdf))
        self.pdf_sampler.jitter_override = draws[1]
        importance = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=self.num_samples_importance)
        self.outside_sampler.jitter_override = draws[2]
        outside = self.outside_sampler(ray_bundle, num_samples=self.num_samples_outside)
        m_bins, _, m_starts, m_ends = self.error_bounded_sampler.merge(ray_bundle, importance.flat_bins, outside.flat_bins)
        mask, z, new_nears, new_fars = self.find_surface(ray_bundle, ray_samples, sdf[..., 0])
        surface_points = ray_bundle.origins[mask] + ray_bundle.directions[mask] * z[mask][:, None]
        if surface_points.shape[0] <= 0:
            surface_points = torch.rand((1024, 3), device=sdf.device) - 0.5  # :1065-1066
        nears, fars = ray_bundle.nears, ray_bundle.fars
        ray_bundle.nears, ray_bundle.fars = new_nears[:, None], new_fars[:, None]
        self.uniform_sampler.jitter_override = draws[3]
        interval = self.uniform_sampler(ray_bundle, num_samples=self.num_samples_interval)
        ray_bundle.nears, ray_bundle.fars = nears, fars
        out = self.merge_ray_samples_in_eculidean(ray_bundle, interval, (m_starts, m_ends))
        return (out, surface_points) if return_surface_points else out

    def merge_ray_samples_in_eculidean(self, ray_bundle: RayBundle, ray_samples_1: RaySamples, ray_samples_2):
        """:1095-1130: sorted union of the euclidean starts, closed by the larger of the two last ends; the merged bins ARE
        euclidean (the reference's TODO: spacing bins = euclidean bins).  Runs on the merge kernel with the identity
        spacing -> euclidean map (near 0, far 1)."""
        s2, e2 = ray_samples_2 if isinstance(ray_samples_2, tuple) else (ray_samples_2.flat_starts, ray_samples_2.flat_ends)
        b1 = torch.cat([ray_samples_1.flat_starts, ray_samples_1.flat_ends[:, -1:]], -1)
        b2 = torch.cat([s2, e2[:, -1:]], -1)
        unit = RayBundle(origins=ray_bundle.origins, directions=ray_bundle.directions, nears=torch.zeros_like(ray_bundle.nears),
                         fars=torch.ones_like(ray_bundle.fars))
        bins, _, starts, ends = self.error_bounded_sampler.merge(unit, b1, b2)
        return ray_bundle.get_ray_samples(
            bin_starts=starts[..., None], bin_ends=ends[..., None], spacing_starts=bins[:, :-1, None], spacing_ends=bins[:, 1:, None],
            spacing_to_euclidean_fn=ray_samples_1.spacing_to_euclidean_fn, flat_bins=bins)


def march_occupancy_grid(origins, directions, t_min, t_max, roi_aabb, binary, step_size: float, capacity: Optional[int] = None,
                         step_dev: Optional[torch.Tensor] = None):
    """nerfacc.cuda.ray_marching as the reference calls it (ray_samplers.py:1474-1484): two native launches (count, write) around an
    exclusive scan.  Returns (packed_info [N,2] int64 = (offset, count), counts [N] int32, ray_indices [P] int64, t_starts [P,1],
    t_ends [P,1]).
    capacity = None: exact-size tensors - ONE host synchronisation, reading the total sample count (the reference's own call returns
    data-dependent shapes as well).
    capacity = C (the bounded form): NO host read.  The arrays have C entries; the first min(total, C) hold the samples, the
    rest a harmless filler (ray 0 at distance 0: a finite point every field kernel accepts) that NO ray's (offset, count) covers, so the
    compositing kernels never touch it; (offsets, counts) are clamped to C on the device.  A sixth value comes back: (n_valid, total) as
    device int64 scalars - the caller masks per-sample reductions with n_valid (models/neus_acc.py) and compares total with C whenever it
    chooses to pay a read (NeuSAccSampler: every `check_every` steps).  step_dev: the step as a device scalar (overrides step_size)."""
    lib = _lib.load()
    n = origins.shape[0]
    dev = origins.device
    kp = _lib.Keep()
    roi = (ctypes.c_float * 6)(*[float(v) for v in roi_aabb.reshape(-1).tolist()])
    occ = binary.contiguous()
    assert occ.dtype == torch.bool and occ.dim() == 3 and occ.shape[0] == occ.shape[1] == occ.shape[2]
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    o, d, tn, tf = kp(origins), kp(directions), kp(t_min.reshape(-1)), kp(t_max.reshape(-1))
    sd = None if step_dev is None else kp(step_dev.detach().reshape(1).float())
    if capacity is None and step_dev is None:
        _lib.check(lib.sdfhip_march_count(o, d, tn, tf, roi, occ.data_ptr(), n, occ.shape[0], float(step_size), counts.data_ptr(),
                                          _lib.stream()), "march_count")
    else:
        _lib.check(lib.sdfhip_march_count_dev(o, d, tn, tf, roi, occ.data_ptr(), n, occ.shape[0], float(step_size), sd, counts.data_ptr(),
                                              _lib.stream()), "march_count_dev")
    ends = torch.cumsum(counts.long(), dim=0)
    offsets = ends - counts.long()
    if capacity is None:
        total = int(ends[-1].item()) if n > 0 else 0
        ray_indices = torch.empty(total, dtype=torch.int64, device=dev)
        t_starts = torch.empty(total, 1, device=dev)
        t_ends = torch.empty(total, 1, device=dev)
        if total > 0:
            if step_dev is None:
                _lib.check(lib.sdfhip_march_write(o, d, tn, tf, roi, occ.data_ptr(), n, occ.shape[0], float(step_size), offsets.data_ptr(),
                                                  ray_indices.data_ptr(), _lib.ptr(t_starts), _lib.ptr(t_ends), _lib.stream()), "march_write")
            else:
                _lib.check(lib.sdfhip_march_write_capped(o, d, tn, tf, roi, occ.data_ptr(), n, occ.shape[0], float(step_size), sd,
                                                         offsets.data_ptr(), -1, ray_indices.data_ptr(), _lib.ptr(t_starts), _lib.ptr(t_ends),
                                                         _lib.stream()), "march_write_capped")
        del kp
        return torch.stack([offsets, counts.long()], dim=-1), counts, ray_indices, t_starts, t_ends
    cap = int(capacity)
    assert cap > 0 and n > 0
    ray_indices = torch.zeros(cap, dtype=torch.int64, device=dev)
    t_starts = torch.zeros(cap, 1, device=dev)
    t_ends = torch.zeros(cap, 1, device=dev)
    _lib.check(lib.sdfhip_march_write_capped(o, d, tn, tf, roi, occ.data_ptr(), n, occ.shape[0], float(step_size), sd, offsets.data_ptr(), cap,
                                             ray_indices.data_ptr(), _lib.ptr(t_starts), _lib.ptr(t_ends), _lib.stream()), "march_write_capped")
    del kp
    total = ends[-1]
    offs_c = offsets.clamp(max=cap)
    counts_c = (ends.clamp(max=cap) - offs_c).to(torch.int32)
    return torch.stack([offs_c, counts_c.long()], dim=-1), counts_c, ray_indices, t_starts, t_ends, (total.clamp(max=cap), total)


def resample_packed(packed_info, counts, t_starts, t_ends, weights, n_samples: int):
    """nerfacc.ray_resampling as the reference calls it (ray_samplers.py:1496-1498) + nerfacc.unpack_info: every ray with samples
    gets n_samples new intervals from the inverse CDF of its weights.  Returns (packed_info, counts, ray_indices, t_starts, t_ends)."""
    lib = _lib.load()
    dev = t_starts.device
    n = counts.shape[0]
    new_counts = (counts > 0).to(torch.int32) * n_samples
    ends = torch.cumsum(new_counts.long(), dim=0)
    offsets = ends - new_counts.long()
    total = int(ends[-1].item()) if n > 0 else 0
    out_s, out_e = torch.empty(total, 1, device=dev), torch.empty(total, 1, device=dev)
    kp = _lib.Keep()
    src_offsets = packed_info[:, 0].contiguous()  # bound to a local: the pointer must outlive the launch
    if total > 0:
        _lib.check(lib.sdfhip_packed_resample(kp(t_starts.reshape(-1)), kp(t_ends.reshape(-1)), kp(weights.reshape(-1).float()),
                                              src_offsets.data_ptr(), counts.data_ptr(), n, n_samples, offsets.data_ptr(),
                                              _lib.ptr(out_s), _lib.ptr(out_e), _lib.stream()), "packed_resample")
    del kp, src_offsets
    ray_indices = torch.repeat_interleave(torch.arange(n, device=dev), new_counts.long())  # nerfacc.unpack_info
    return torch.stack([offsets, new_counts.long()], dim=-1), new_counts, ray_indices, out_s, out_e


class NeuSAccSampler(Sampler):
    """ray_samplers.py:1315-1503: the voxel-surface guided sampler of NeuS-acc.  An occupancy grid over the scene box, pruned every
    `steps_per_grid_update` steps from the SDF at the voxel centres (update_binary_grid), drives a fixed-step march that only
    keeps samples in occupied voxels (packed samples: one flat array for all rays); until the first grid update the model runs on
    the NeuS sampler.  With importance_sampling (off by default, :1326) the marched samples are re-drawn, 16 per ray, from the pdf of
    their own alpha-composited weights (nerfacc.ray_resampling -> sdfhip_packed_resample)."""

    def __init__(self, aabb, neus_sampler: Optional[NeuSSampler] = None, resolution: int = 128, num_samples: int = 8,
                 num_samples_importance: int = 16, num_samples_boundary: int = 10, steps_warpup: int = 2000,
                 steps_per_grid_update: int = 1000, importance_sampling: bool = False, local_rank: int = 0,
                 single_jitter: bool = False, bounded: bool = False, capacity_slack: float = 1.06, check_every: int = 50) -> None:
        """bounded (round 6; not in the reference): the packed arrays are sized by a BOUND instead of the exact sample
        count, so that a training step contains no device -> host read and the host can enqueue ahead.  The bound is measured by one exact
        call after every occupancy-grid update (x capacity_slack, in multiples of 1024 samples: every filler sample costs a field evaluation - at 1.3 the
        bench leg ran 22 % slower than the exact form it replaces, at 1.06 the sample count of 2048 random rays, which moves by ~1 % from step to
        step, stays inside) and re-checked every `check_every` steps
        against the largest count seen (one read per check); samples beyond it would be dropped - `overflowed_steps` counts the steps
        where that happened.  Same samples, same weights, same losses as the exact form whenever the bound holds (the tail of the arrays is
        a filler no ray covers and the eikonal mean is taken over the valid samples: models/neus_acc.py)."""
        super().__init__()
        self.bounded, self.capacity_slack, self.check_every = bool(bounded), float(capacity_slack), int(check_every)
        self._cap: Optional[int] = None
        self._totals: list = []
        self.overflowed_steps = 0
        self.packed_valid: Optional[torch.Tensor] = None  # device int64 scalar: valid packed samples of the last call (bounded form), else None
        self._step_dev: Optional[torch.Tensor] = None
        self._step_host: Optional[float] = None
        self.resolution, self.num_samples, self.num_samples_importance = resolution, num_samples, num_samples_importance
        self.num_samples_boundary, self.single_jitter, self.importance_sampling = num_samples_boundary, single_jitter, importance_sampling
        self.steps_warpup, self.steps_per_grid_update, self.local_rank = steps_warpup, steps_per_grid_update, local_rank
        self.step_size = 0.01 / 5.0
        self.alpha_thres = 0.001
        assert aabb[0, 0] == aabb[0, 1] and aabb[0, 0] == aabb[0, 2]  # cubic boxes only (:1347-1349)
        assert aabb[1, 0] == aabb[1, 1] and aabb[1, 0] == aabb[1, 2]
        self.grid_size = resolution
        self.voxel_size = float(aabb[1, 0] - aabb[0, 0]) / self.grid_size
        self.neus_sampler = neus_sampler
        self.register_buffer("aabb", aabb.clone().float(), persistent=False)
        self.register_buffer("_binary", torch.ones((self.grid_size,) * 3, dtype=torch.bool))
        self.register_buffer("_update_counter", torch.zeros(1, dtype=torch.int32))
        self._updates_host: Optional[int] = 0  # host mirror of _update_counter (None: unknown, e.g. after a checkpoint load)
        lo, hi = float(aabb[0, 0]) + self.voxel_size / 2.0, float(aabb[1, 0]) - self.voxel_size / 2.0
        off = torch.linspace(lo, hi, self.grid_size)
        x, y, z = torch.meshgrid(off, off, off, indexing="ij")
        self.register_buffer("cube_coordinate", torch.stack([x, y, z], dim=-1).reshape(-1, 3))  # :1362-1377

    @property
    def step_size(self) -> float:
        """The march step as a host float (ray_samplers.py:1345, 1379-1382).  In the bounded form update_step_size keeps it on the device; the
        host value is read back only when somebody asks (update_binary_grid every steps_per_grid_update steps, metrics)."""
        if self._step_host is None and self._step_dev is not None:
            self._step_host = float(self._step_dev.item())
        return self._step_host

    @step_size.setter
    def step_size(self, v: float) -> None:
        self._step_host, self._step_dev = float(v), None

    def update_step_size(self, step, inv_s=None):
        """:1379-1382: step = 14 / inv_s / 16, every iteration, from the TRAINED variance.  The reference reads the scalar back
        (`.item()`); the bounded form leaves it on the device (the march kernels take a device step: sdfhip_march_count_dev)."""
        assert inv_s is not None
        s = inv_s()
        if self.bounded and torch.is_tensor(s) and s.is_cuda:
            self._step_dev, self._step_host = (14.0 / s.detach().reshape(1).float() / 16).contiguous(), None
        else:
            self.step_size = 14.0 / float(s) / 16

    @torch.no_grad()
    def update_binary_grid(self, step, sdf_fn=None, inv_s=None):
        """:1383-1432: voxels whose |sdf| minus the half diagonal still gives alpha > 1e-3 stay occupied; pruned voxels never
        come back.  sdf_fn is the geometry network on explicit points (one native call per 100 000 voxels, like the reference)."""
        assert sdf_fn is not None and inv_s is not None
        if step >= self.steps_warpup and step % self.steps_per_grid_update == 0:
            mask = self._binary.reshape(-1)
            occupied_voxel = self.cube_coordinate[mask]
            sdf = torch.cat([sdf_fn(p) for p in torch.split(occupied_voxel, 100000, dim=0)], dim=0) if occupied_voxel.shape[0] else \
                occupied_voxel.new_zeros(0)
            bound = self.voxel_size * (3 ** 0.5) / 2.0
            sdf = torch.maximum(sdf.abs() - bound, torch.zeros_like(sdf))
            s = inv_s()
            prev_cdf = torch.sigmoid((sdf + self.step_size * 0.5) * s)
            next_cdf = torch.sigmoid((sdf - self.step_size * 0.5) * s)
            alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
            mask[mask.clone()] = alpha > self.alpha_thres
            self._binary = mask.reshape([self.grid_size] * 3).contiguous()
            self._cap, self._totals = None, []  # bounded form: the occupancy changed, the next call measures the bound again
            self._update_counter += 1
            if self._updates_host is not None:
                self._updates_host += 1

    def check_capacity(self) -> None:
        """Bounded form: ONE device -> host read for the last `check_every` (or fewer) steps - the largest exact sample count among them.
        Counts the steps that overflowed the bound (their last samples were dropped) and widens the bound when the count comes within 2 %."""
        if not self._totals:
            return
        totals = torch.stack(self._totals).cpu().tolist()
        self._totals = []
        self.overflowed_steps += sum(1 for t in totals if t > self._cap)
        m = max(totals)
        if m > 0.98 * self._cap:
            self._cap = max(1024, -(-int(self.capacity_slack * m) // 1024) * 1024)

    def num_grid_updates(self) -> int:
        """The reference reads `_update_counter.item()` in every forward (ray_samplers.py:1467, models/neus_acc.py:93): a device -> host
        synchronisation per call.  The counter only changes in update_binary_grid and on a checkpoint load, so a host mirror answers."""
        if self._updates_host is None:
            self._updates_host = int(self._update_counter.item())
        return self._updates_host

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._updates_host = None  # re-read from the loaded buffer on next use

    def create_ray_samples_from_ray_indices(self, ray_bundle: RayBundle, ray_indices, t_starts, t_ends) -> RaySamples:
        """:1434-1455.  Packed samples are [P,1] here (every sample its own one-sample ray), the layout the field kernels take."""
        packed = RayBundle(origins=ray_bundle.origins[ray_indices], directions=ray_bundle.directions[ray_indices],
                           pixel_area=torch.ones(ray_indices.shape[0], 1, device=t_starts.device),
                           camera_indices=None if ray_bundle.camera_indices is None else ray_bundle.camera_indices[ray_indices])
        return packed.get_ray_samples(t_starts[:, None, :], t_ends[:, None, :])

    @torch.no_grad()
    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, sdf_fn: Optional[Callable] = None,
                             alpha_fn: Optional[Callable] = None):
        """:1457-1503.  After the first grid update: (ray_samples [P,1], ray_indices [P]); `packed_info` / `counts` of the call are
        kept on the sampler for the compositing kernels."""
        assert ray_bundle is not None and sdf_fn is not None
        if self.num_grid_updates() <= 0:
            return self.neus_sampler(ray_bundle, sdf_fn=sdf_fn)
        self.packed_valid = None
        args = (ray_bundle.origins, ray_bundle.directions, ray_bundle.nears[:, 0], ray_bundle.fars[:, 0], self.aabb, self._binary)
        if self.bounded and self._cap is not None:
            assert not self.importance_sampling, "bounded packed sampling is built for the plain march (importance_sampling=False, the reference's default)"
            info, counts, ray_indices, t_starts, t_ends, (n_valid, total) = march_occupancy_grid(
                *args, self._step_host if self._step_dev is None else 1.0, capacity=self._cap, step_dev=self._step_dev)
            self.packed_valid = n_valid
            self._totals.append(total)
            if len(self._totals) >= self.check_every:
                self.check_capacity()
        else:
            info, counts, ray_indices, t_starts, t_ends = march_occupancy_grid(*args, self.step_size)
            if self.bounded:  # the exact call after a grid update sizes the bound
                self._cap = max(1024, -(-int(self.capacity_slack * ray_indices.shape[0]) // 1024) * 1024)
        ray_samples = self.create_ray_samples_from_ray_indices(ray_bundle, ray_indices, t_starts, t_ends)
        if self.importance_sampling and ray_samples.shape[0] > 0:  # :1489-1500
            from sdfstudio_amd.model_components.renderers import render_weight_from_alpha

            assert alpha_fn is not None
            alphas = alpha_fn(ray_samples)[:, 0, :]
            weights = render_weight_from_alpha(alphas, info, counts)
            info, counts, ray_indices, t_starts, t_ends = resample_packed(info, counts, t_starts, t_ends, weights[:, 0], 16)
            ray_samples = self.create_ray_samples_from_ray_indices(ray_bundle, ray_indices, t_starts, t_ends)
        self.packed_info, self.packed_counts = info, counts
        return ray_samples, ray_indices
