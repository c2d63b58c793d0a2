"""Losses that close the NeuS-facto training step.

interlevel_loss_zip restates nerfstudio/model_components/losses.py:116-172 (Zip-NeRF proposal loss) on flat
``[N,S+1]`` spacing-domain bins.  Only the proposal weights carry gradient (the field's weights are detached, :134).
Each proposal level is ONE kernel launch (sdfhip_interlevel_terms: one wavefront per ray, the blurred histogram built by
merging the two sorted knot sequences instead of sorting them) plus a mean; the reference's formulation (sort, gathers, cumsums,
searchsorted: ~40 launches) lives in oracle/sdf_path.py as the checker.  There is no CPU path: CPU tensors raise.  The mono-prior
losses are small torch reductions over rendered per-ray outputs.
"""
from typing import List

import torch

from sdfstudio_amd import _lib


class _InterlevelLevel(torch.autograd.Function):
    """mean over rays and samples of clip(w_gt - wp, 0)^2 / (wp + 1e-5) for one proposal level (losses.py:156-171)."""

    @staticmethod
    def forward(ctx, wp, c, w, cp, radius):
        lib = _lib.load()
        n, s_p = wp.shape
        s = w.shape[1]
        kp = _lib.Keep()
        term = torch.empty(n, s_p, device=wp.device)
        dterm = torch.empty(n, s_p, device=wp.device)
        _lib.check(lib.sdfhip_interlevel_terms(kp(c), kp(w), kp(cp), kp(wp.detach()), n, s, s_p, float(radius), _lib.ptr(term),
                                               _lib.ptr(dterm), None, _lib.stream()), "interlevel_terms")
        del kp
        ctx.save_for_backward(dterm)
        return term.mean()

    @staticmethod
    def backward(ctx, g):
        (dterm,) = ctx.saved_tensors
        return dterm * (g / dterm.numel()), None, None, None, None


def interlevel_loss_zip(weights_list: List[torch.Tensor], bins_list: List[torch.Tensor]) -> torch.Tensor:
    """weights_list[i]: [N,S_i] (last = field weights), bins_list[i]: [N,S_i+1] spacing bins (last = field bins)."""
    c = bins_list[-1].detach()
    w = weights_list[-1].detach()
    total = 0.0
    for cp, wp, radius in zip(bins_list[:-1], weights_list[:-1], (0.03, 0.003)):
        total = total + _InterlevelLevel.apply(wp, c, w, cp.detach(), radius)
    return total


def monosdf_normal_loss(normal_pred: torch.Tensor, normal_gt: torch.Tensor) -> torch.Tensor:
    """losses.py:264-275: L1 + cosine between the rendered normal and the monocular normal prior."""
    n_gt = torch.nn.functional.normalize(normal_gt, p=2, dim=-1)
    n_pr = torch.nn.functional.normalize(normal_pred, p=2, dim=-1)
    return torch.abs(n_pr - n_gt).sum(dim=-1).mean() + (1.0 - torch.sum(n_pr * n_gt, dim=-1)).mean()


def scale_and_shift_invariant_loss(prediction: torch.Tensor, target: torch.Tensor, mask: torch.Tensor, alpha: float = 0.5,
                                   scales: int = 1) -> torch.Tensor:
    """losses.py:278-409 ScaleAndShiftInvariantLoss (MiDaS), batch-based reduction, as used for the MonoSDF depth prior
    (base_surface_model.py:227,427-437).  prediction / target / mask: [B,H,W].

    Per image: least-squares scale s and shift t of the prediction against the target (closed form 2x2 system, zero when the
    system is singular), then MSE(s p + t, target) / 2 + alpha * sum over `scales` of the masked gradient-matching term."""
    m = mask.to(prediction.dtype)
    a00 = (m * prediction * prediction).sum((1, 2))
    a01 = (m * prediction).sum((1, 2))
    a11 = m.sum((1, 2))
    b0 = (m * prediction * target).sum((1, 2))
    b1 = (m * target).sum((1, 2))
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    scale = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(det))
    shift = torch.where(ok, (-a01 * b0 + a00 * b1) / safe, torch.zeros_like(det))
    ssi = scale.view(-1, 1, 1) * prediction + shift.view(-1, 1, 1)

    def batch_mean(image_loss, count):
        total = count.sum()
        return image_loss.sum() / total if total != 0 else image_loss.sum() * 0.0

    res = ssi - target
    total = batch_mean((m * res * res).sum((1, 2)), 2 * a11)
    if alpha > 0:
        reg = 0.0
        for k in range(scales):
            st = 2 ** k
            p_, t_, m_ = ssi[:, ::st, ::st], target[:, ::st, ::st], m[:, ::st, ::st]
            diff = m_ * (p_ - t_)
            gx = (m_[:, :, 1:] * m_[:, :, :-1]) * torch.abs(diff[:, :, 1:] - diff[:, :, :-1])
            gy = (m_[:, 1:, :] * m_[:, :-1, :]) * torch.abs(diff[:, 1:, :] - diff[:, :-1, :])
            reg = reg + batch_mean(gx.sum((1, 2)) + gy.sum((1, 2)), m_.sum((1, 2)))
        total = total + alpha * reg
    return total


def monosdf_depth_loss(depth_pred: torch.Tensor, depth_gt: torch.Tensor) -> torch.Tensor:
    """base_surface_model.py:427-437: the ray batch is viewed as a 32 x (N/32) patch, the prior is rescaled (x 50 + 0.5)."""
    mask = torch.ones_like(depth_gt).reshape(1, 32, -1).bool()
    return scale_and_shift_invariant_loss(depth_pred.reshape(1, 32, -1), (depth_gt * 50 + 0.5).reshape(1, 32, -1), mask, 0.5, 1)
