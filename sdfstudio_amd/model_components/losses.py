"""Losses that close the NeuS-facto training step (kept as PyTorch device ops this round).

interlevel_loss_zip restates nerfstudio/model_components/losses.py:116-172 (Zip-NeRF proposal loss) on flat
``[N,S+1]`` spacing-domain bins.  Only the proposal weights carry gradient (the field's weights are detached, :134).
"""
from typing import List

import torch


def _blurred_step(bins: torch.Tensor, heights: torch.Tensor, radius: float):
    """losses.py:116-128 blur_stepfun: piecewise-linear blur of a step function with a box of half-width `radius`."""
    knots = torch.cat([bins - radius, bins + radius], dim=-1)
    knots_sorted, order = torch.sort(knots, dim=-1)
    zero = torch.zeros_like(heights[:, :1])
    jumps = (torch.cat([heights, zero], dim=-1) - torch.cat([zero, heights], dim=-1)) / (2 * radius)
    slopes = torch.gather(torch.cat([jumps, -jumps], dim=-1), -1, order[:, :-1])
    values = torch.cumsum((knots_sorted[:, 1:] - knots_sorted[:, :-1]) * torch.cumsum(slopes, dim=-1), dim=-1)
    return knots_sorted, torch.cat([zero, values], dim=-1)


def interlevel_loss_zip(weights_list: List[torch.Tensor], bins_list: List[torch.Tensor]) -> torch.Tensor:
    """weights_list[i]: [N,S_i] (last = field weights), bins_list[i]: [N,S_i+1] spacing bins (last = field bins)."""
    c = bins_list[-1].detach()
    w = weights_list[-1].detach()
    w_norm = w / (c[:, 1:] - c[:, :-1])
    total = 0.0
    for cp, wp, radius in zip(bins_list[:-1], weights_list[:-1], (0.03, 0.003)):
        xr, yr = _blurred_step(c, w_norm, radius)
        yr = torch.clip(yr, min=0)
        area = torch.cumsum((yr[:, 1:] + yr[:, :-1]) * 0.5 * (xr[:, 1:] - xr[:, :-1]), dim=-1)
        area = torch.cat([torch.zeros_like(area[:, :1]), area], dim=-1)
        cp = cp.detach().contiguous()
        idx = torch.searchsorted(xr, cp, side="right")
        top = xr.shape[-1] - 1
        lo, hi = torch.clamp(idx - 1, 0, top), torch.clamp(idx, 0, top)
        x0, x1 = torch.gather(xr, -1, lo), torch.gather(xr, -1, hi)
        a0, a1 = torch.gather(area, -1, lo), torch.gather(area, -1, hi)
        t = torch.clip(torch.nan_to_num((cp - x0) / (x1 - x0), 0), 0, 1)
        cum = a0 + t * (a1 - a0)
        w_target = cum[:, 1:] - cum[:, :-1]
        total = total + torch.mean(torch.clip(w_target - wp, min=0) ** 2 / (wp + 1e-5))
    return total
