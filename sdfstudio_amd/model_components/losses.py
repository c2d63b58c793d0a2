"""Losses that close the NeuS-facto training step.

interlevel_loss_zip restates nerfstudio/model_components/losses.py:116-172 (Zip-NeRF proposal loss) on flat
``[N,S+1]`` spacing-domain bins.  Only the proposal weights carry gradient (the field's weights are detached, :134).
Each proposal level is ONE kernel launch (sdfhip_interlevel_terms: one wavefront per ray, the blurred histogram built by
merging the two sorted knot sequences instead of sorting them) plus a mean; the reference's formulation (sort, gathers, cumsums,
searchsorted: ~40 launches) lives in oracle/sdf_path.py as the checker.  There is no CPU path: CPU tensors raise.

surface_losses: the scalar losses behind the renderer - L1 colour (base_surface_model.py:402), eikonal (:406), curvature
(neus_facto.py:313-325), MonoSDF normal (losses.py:264-275) - as one fused operator: two launches forward (per-block partial sums,
deterministic finish), one elementwise launch backward (sdfhip_surface_loss_forward / _backward), instead of ~25 reductions and
elementwise launches.  The MonoSDF scale-and-shift depth loss (a 2 x 2 least-squares fit + gradient matching on the ray batch viewed
as an image, losses.py:278-409; BASELINE config 4) and the foreground-mask BCE are one native launch each way as well
(sdfhip_mono_depth_loss_*, sdfhip_fg_mask_loss_*); scale_and_shift_invariant_loss below is the general statement (any mask, any number of
scales) for callers outside the surface models.
"""
import ctypes
import math
from typing import Dict, List, Optional

import torch

from sdfstudio_amd import _lib


class _SurfaceLosses(torch.autograd.Function):
    """(rgb, eik_grad, sdf, taps, n_pred) -> (rgb_loss, eikonal_loss, curvature_loss, normal_loss), each already multiplied by its
    weight; absent inputs are None and their loss is 0."""

    @staticmethod
    def forward(ctx, rgb, grad, sdf, taps, n_pred, image, n_gt, delta, mults):
        lib = _lib.load()
        if not rgb.is_cuda:
            raise _lib.SdfHipError("surface_losses runs on the sdfhip kernels: HIP device tensors required (no CPU fallback)")
        dev = rgb.device
        n = rgb.shape[0]
        P = grad.numel() // 3 if grad is not None else (sdf.numel() if sdf is not None else 0)
        scale = (ctypes.c_float * 4)(mults[0] / (3.0 * n), (mults[1] / P) if grad is not None else 0.0,
                                     (mults[2] / (3.0 * P)) if taps is not None else 0.0, (mults[3] / n) if n_pred is not None else 0.0)
        ws = torch.empty(lib.sdfhip_surface_loss_workspace_floats(), device=dev)
        loss4 = torch.empty(4, device=dev)
        # the marshalled inputs, in the ABI's order: rgb, image, grad, sdf, taps, n_pred, n_gt (None where a loss is switched off)
        ts = [rgb.detach().float().contiguous(), image.float().contiguous(), None if grad is None else grad.detach().contiguous(),
              None if sdf is None else sdf.detach().contiguous(), None if taps is None else taps.detach().contiguous(),
              None if n_pred is None else n_pred.detach().contiguous(), None if n_gt is None else n_gt.float().contiguous()]
        _lib.check(lib.sdfhip_surface_loss_forward(*_SurfaceLosses._args(ts, n, float(delta), P), scale, _lib.ptr(ws), _lib.ptr(loss4), _lib.stream()),
                   "surface_loss_forward")
        # saved THROUGH autograd (not as raw pointers): an in-place change of rgb / eik_grad / sdf between forward and backward trips the
        # version-counter check instead of silently producing wrong gradients, and the tensors are released with the graph
        ctx.present = [t is not None for t in ts]
        ctx.save_for_backward(*[t for t in ts if t is not None])
        ctx.scale, ctx.n, ctx.delta, ctx.P = scale, n, float(delta), P
        ctx.shapes = (rgb.shape, None if grad is None else grad.shape, None if sdf is None else sdf.shape, None if taps is None else taps.shape,
                      None if n_pred is None else n_pred.shape)
        return loss4[0:1].view(()), loss4[1:2].view(()), loss4[2:3].view(()), loss4[3:4].view(())

    @staticmethod
    def _args(ts, n, delta, P):
        """(rgb, image, n, grad, sdf, taps, delta, P, n_pred, n_gt) as the ABI takes them; `ts` stays alive in the caller until the launch."""
        return (_lib.ptr(ts[0]), _lib.ptr(ts[1]), n, _lib.ptr(ts[2]), _lib.ptr(ts[3]), _lib.ptr(ts[4]), delta, P, _lib.ptr(ts[5]), _lib.ptr(ts[6]))

    @staticmethod
    def backward(ctx, *lbar):
        lib = _lib.load()
        saved = list(ctx.saved_tensors)
        ts = [saved.pop(0) if here else None for here in ctx.present]
        dev = ts[0].device
        shp = ctx.shapes
        need = ctx.needs_input_grad
        lb = [None if t is None else t.contiguous().float() for t in lbar]
        outs = [torch.empty(shp[i], device=dev) if (shp[i] is not None and need[i]) else None for i in range(5)]
        if outs[2] is not None and shp[3] is None:
            outs[2] = None  # the kernel writes sdf_bar in its curvature pass only: without taps the sdf has no gradient from these losses
        # the curvature stencil differentiates sdf and its six taps in one pass: whichever of the two is asked for, both buffers exist
        # (the other one is scratch) - a sdf that requires grad next to detached taps used to lose its curvature gradient
        curv = shp[3] is not None and (need[2] or need[3])
        sdf_bar = outs[2] if outs[2] is not None else (torch.empty(shp[2], device=dev) if curv else None)
        taps_bar = outs[3] if outs[3] is not None else (torch.empty(shp[3], device=dev) if curv else None)
        _lib.check(lib.sdfhip_surface_loss_backward(*_SurfaceLosses._args(ts, ctx.n, ctx.delta, ctx.P), ctx.scale, _lib.ptr_array(lb), _lib.ptr(outs[0]),
                                                    _lib.ptr(outs[1]), _lib.ptr(sdf_bar), _lib.ptr(taps_bar), _lib.ptr(outs[4]), _lib.stream()),
                   "surface_loss_backward")
        del lb, ts
        return outs[0], outs[1], sdf_bar if need[2] else None, taps_bar if need[3] else None, outs[4], None, None, None, None


def surface_losses(rgb: torch.Tensor, image: torch.Tensor, eik_grad: Optional[torch.Tensor] = None, eikonal_mult: float = 0.0,
                   sdf: Optional[torch.Tensor] = None, sampled_sdf: Optional[torch.Tensor] = None, delta: float = 1.0, curvature_mult: float = 0.0,
                   normal_pred: Optional[torch.Tensor] = None, normal_gt: Optional[torch.Tensor] = None, normal_mult: float = 0.0) -> Dict[str, torch.Tensor]:
    """The fused scalar losses of SurfaceModel.get_loss_dict (base_surface_model.py:399-424, neus_facto.py:312-325).  rgb / image
    [N,3]; eik_grad [N,S,3]; sdf [N,S,1] + sampled_sdf [N,S,6] (numerical-gradient taps); normal_pred / normal_gt [N,3].  Returns
    the entries that are switched on, multiplied by their weights: rgb_loss always, eikonal_loss / curvature_loss / normal_loss."""
    use_curv = sampled_sdf is not None and curvature_mult > 0.0
    use_nrm = normal_pred is not None and normal_mult > 0.0
    l_rgb, l_eik, l_cur, l_nrm = _SurfaceLosses.apply(
        rgb.contiguous(), None if eik_grad is None else eik_grad.contiguous(), sdf.contiguous() if use_curv else None,
        sampled_sdf.contiguous() if use_curv else None, normal_pred.contiguous() if use_nrm else None, image, normal_gt if use_nrm else None,
        delta, (1.0, eikonal_mult, curvature_mult, normal_mult))
    out = {"rgb_loss": l_rgb}
    if eik_grad is not None:
        out["eikonal_loss"] = l_eik
    if use_curv:
        out["curvature_loss"] = l_cur
    if use_nrm:
        out["normal_loss"] = l_nrm
    return out


class _Interlevel(torch.autograd.Function):
    """interlevel_loss_zip (losses.py:116-172) as ONE autograd node over all proposal levels: sum over levels of the mean over rays and
    samples of clip(w_gt - wp, 0)^2 / (wp + 1e-5).  One native launch per level writes the terms and their derivatives into slices of two
    flat buffers; the means of all levels are ONE weighted sum with a cached vector of 1 / numel per element, and the backward is one
    multiply of the (pre-scaled) derivatives by the incoming gradient - four ATen launches where a node per level, its mean and the
    running sum took ten."""

    _scale_cache = {}

    @staticmethod
    def forward(ctx, c, w, radii, *levels):  # levels = cp_0, wp_0, cp_1, wp_1, ...
        lib = _lib.load()
        n, s = w.shape
        dev = w.device
        shapes = [tuple(levels[2 * i + 1].shape) for i in range(len(levels) // 2)]
        sizes = [a * b for a, b in shapes]
        key = (tuple(sizes), str(dev))
        scale = _Interlevel._scale_cache.get(key)
        if scale is None:
            if len(_Interlevel._scale_cache) > 16:
                _Interlevel._scale_cache.clear()
            scale = torch.cat([torch.full((m,), 1.0 / m, device=dev) for m in sizes])
            _Interlevel._scale_cache[key] = scale
        term = torch.empty(sum(sizes), device=dev)
        dterm = torch.empty(sum(sizes), device=dev)
        kp = _lib.Keep()
        c_p, w_p = kp(c), kp(w)
        off = 0
        for i, (m, (_, s_p)) in enumerate(zip(sizes, shapes)):
            cp, wp = levels[2 * i], levels[2 * i + 1]
            _lib.check(lib.sdfhip_interlevel_terms(c_p, w_p, kp(cp), kp(wp.detach()), n, s, s_p, float(radii[i]), _lib.ptr(term[off:off + m]),
                                                   _lib.ptr(dterm[off:off + m]), None, _lib.stream()), "interlevel_terms")
            off += m
        del kp
        ctx.save_for_backward(dterm.mul_(scale))  # d (sum of means) / d wp, all levels
        ctx.shapes = shapes
        return term.mul_(scale).sum()  # (not torch.dot: that is a rocBLAS call with its own workspace and launches)

    @staticmethod
    def backward(ctx, g):
        (dscaled,) = ctx.saved_tensors
        out = dscaled * g
        grads, off = [None, None, None], 0
        for a, b in ctx.shapes:
            grads += [None, out[off:off + a * b].view(a, b)]
            off += a * b
        return tuple(grads)


def interlevel_loss_zip(weights_list: List[torch.Tensor], bins_list: List[torch.Tensor]) -> torch.Tensor:
    """weights_list[i]: [N,S_i] (last = field weights), bins_list[i]: [N,S_i+1] spacing bins (last = field bins)."""
    c = bins_list[-1].detach()
    w = weights_list[-1].detach()
    radii = (0.03, 0.003)  # losses.py:138: one blur half-width per proposal level
    levels = []
    for cp, wp in zip(bins_list[:-1][:len(radii)], weights_list[:-1][:len(radii)]):
        levels += [cp.detach(), wp]
    if not levels:  # no proposal level (num_proposal_iterations = 0): the reference's loop (losses.py:160-172) returns 0.0
        return w.new_zeros(())
    if not w.is_cuda:
        raise _lib.SdfHipError("interlevel_loss_zip runs on the sdfhip kernels: HIP device tensors required (no CPU fallback)")
    return _Interlevel.apply(c, w, radii[:len(levels) // 2], *levels)


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


class _MonoDepthLoss(torch.autograd.Function):
    """ScaleAndShiftInvariantLoss(alpha = 0.5, scales = 1) on the ray batch viewed as one 32-row image with an all-ones mask: one native
    launch forward (fit, loss and the sums the backward needs), one backward - the gradient goes THROUGH the scale / shift fit as it
    does under autograd in the reference."""

    @staticmethod
    def forward(ctx, depth_pred, depth_gt, rows, gt_scale, gt_shift, alpha):
        lib = _lib.load()
        p = depth_pred.detach().reshape(-1).contiguous().float()
        g = depth_gt.detach().reshape(-1).contiguous().float()
        loss = torch.empty(1, device=p.device)
        state = torch.empty(10, device=p.device)
        _lib.check(lib.sdfhip_mono_depth_loss_forward(_lib.ptr(p), _lib.ptr(g), p.numel(), int(rows), float(gt_scale), float(gt_shift),
                                                      float(alpha), _lib.ptr(loss), _lib.ptr(state), _lib.stream()), "mono_depth_loss_forward")
        ctx.save_for_backward(p, g, state)
        ctx.cfg, ctx.shape = (int(rows), float(gt_scale), float(gt_shift), float(alpha)), depth_pred.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, lbar):
        lib = _lib.load()
        p, g, state = ctx.saved_tensors
        rows, gt_scale, gt_shift, alpha = ctx.cfg
        lb = lbar.reshape(1).contiguous().float()
        out = torch.empty_like(p)
        _lib.check(lib.sdfhip_mono_depth_loss_backward(_lib.ptr(p), _lib.ptr(g), p.numel(), rows, gt_scale, gt_shift, alpha, _lib.ptr(state),
                                                       _lib.ptr(lb), _lib.ptr(out), _lib.stream()), "mono_depth_loss_backward")
        return out.view(ctx.shape), None, None, None, None, None


def monosdf_depth_loss(depth_pred: torch.Tensor, depth_gt: torch.Tensor) -> torch.Tensor:
    """base_surface_model.py:427-437: the ray batch is viewed as a 32 x (N/32) patch, the prior is rescaled (x 50 + 0.5).  On the device:
    the fused operator (sdfhip_mono_depth_loss_*); CPU tensors (host-side checks against the reference's class) take the statement."""
    if depth_pred.is_cuda:
        return _MonoDepthLoss.apply(depth_pred, depth_gt, 32, 50.0, 0.5, 0.5)
    mask = torch.ones_like(depth_gt).reshape(1, 32, -1).bool()
    return scale_and_shift_invariant_loss(depth_pred.reshape(1, 32, -1), (depth_gt * 50 + 0.5).reshape(1, 32, -1), mask, 0.5, 1)


class _FgMaskLoss(torch.autograd.Function):
    """mult * binary_cross_entropy(clip(acc, 1e-3, 1 - 1e-3), label) over the rays: one native launch each way."""

    @staticmethod
    def forward(ctx, acc, label, mult):
        lib = _lib.load()
        a = acc.detach().reshape(-1).contiguous().float()
        y = label.detach().reshape(-1).contiguous().float()
        loss = torch.empty(1, device=a.device)
        _lib.check(lib.sdfhip_fg_mask_loss_forward(_lib.ptr(a), _lib.ptr(y), a.numel(), float(mult), _lib.ptr(loss), _lib.stream()),
                   "fg_mask_loss_forward")
        ctx.save_for_backward(a, y)
        ctx.mult, ctx.shape = float(mult), acc.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, lbar):
        lib = _lib.load()
        a, y = ctx.saved_tensors
        lb = lbar.reshape(1).contiguous().float()
        out = torch.empty_like(a)
        _lib.check(lib.sdfhip_fg_mask_loss_backward(_lib.ptr(a), _lib.ptr(y), a.numel(), ctx.mult, _lib.ptr(lb), _lib.ptr(out), _lib.stream()),
                   "fg_mask_loss_backward")
        return out.view(ctx.shape), None, None


def fg_mask_loss(weights_sum: torch.Tensor, fg_label: torch.Tensor, mult: float) -> torch.Tensor:
    """base_surface_model.py:415-420: mult * BCE(clip(sum of the rendering weights per ray, 1e-3, 1 - 1e-3), foreground mask)."""
    if weights_sum.is_cuda:
        return _FgMaskLoss.apply(weights_sum, fg_label, mult)
    return torch.nn.functional.binary_cross_entropy(weights_sum.clip(1e-3, 1.0 - 1e-3), fg_label) * mult


class _SensorDepthLoss(torch.autograd.Function):
    """SensorDepthLoss (model_components/losses.py:628-676) as one native operator: {l1, free space, sdf} from the rendered depth, the
    sensor depth and the field's per-sample sdf; gradients for the rendered depth and the sdf values (sdfhip_sensor_depth_loss_*)."""

    @staticmethod
    def forward(ctx, depth_pred, depth_gt, sdf, starts, directions_norm, truncation):
        lib = _lib.load()
        n, s = sdf.shape[0], sdf.shape[1]
        dp = depth_pred.detach().reshape(-1).contiguous().float()
        dg = depth_gt.detach().reshape(-1).contiguous().float()
        x = sdf.detach().reshape(n, s).contiguous().float()
        st = starts.detach().reshape(n, s).contiguous().float()
        dn = None if directions_norm is None else directions_norm.detach().reshape(-1).contiguous().float()
        if n == 0 or s == 0:
            raise _lib.SdfHipError("sensor_depth_loss: empty batch")
        assert dp.numel() == n and dg.numel() == n and (dn is None or dn.numel() == n)
        ws = torch.empty(int(lib.sdfhip_sensor_depth_loss_workspace_size()), dtype=torch.uint8, device=x.device)
        losses, state = torch.empty(3, device=x.device), torch.empty(4, device=x.device)
        _lib.check(lib.sdfhip_sensor_depth_loss_forward(_lib.ptr(dp), _lib.ptr(dg), _lib.ptr(x), _lib.ptr(st), _lib.ptr(dn), n, s, float(truncation),
                                                        _lib.rawptr(ws), _lib.ptr(losses), _lib.ptr(state), _lib.stream()), "sensor_depth_loss_forward")
        ctx.save_for_backward(dp, dg, x, st, state, *(() if dn is None else (dn,)))
        ctx.t, ctx.shapes = float(truncation), (depth_pred.shape, sdf.shape)
        return losses

    @staticmethod
    def backward(ctx, lbar):
        lib = _lib.load()
        dp, dg, x, st, state, *rest = ctx.saved_tensors
        dn = rest[0] if rest else None
        lb = lbar.reshape(3).contiguous().float()
        sdf_bar, depth_bar = torch.empty_like(x), torch.empty_like(dp)
        _lib.check(lib.sdfhip_sensor_depth_loss_backward(_lib.ptr(dp), _lib.ptr(dg), _lib.ptr(x), _lib.ptr(st), _lib.ptr(dn), x.shape[0], x.shape[1], ctx.t,
                                                         _lib.ptr(state), _lib.ptr(lb), _lib.ptr(sdf_bar), _lib.ptr(depth_bar), _lib.stream()),
                   "sensor_depth_loss_backward")
        return depth_bar.view(ctx.shapes[0]), None, sdf_bar.view(ctx.shapes[1]), None, None, None


def sensor_depth_loss(depth_pred: torch.Tensor, depth_gt: torch.Tensor, pred_sdf: torch.Tensor, starts: torch.Tensor,
                      directions_norm: Optional[torch.Tensor], truncation: float):
    """SensorDepthLoss.forward (model_components/losses.py:635-676) on explicit tensors: depth_pred [N, 1] (outputs["depth"]), depth_gt [N]
    (batch["sensor_depth"]; <= 0 = no measurement), pred_sdf [N, S] (field_outputs[SDF][..., 0]), starts [N, S] (ray_samples.frustums.starts[..., 0]),
    directions_norm [N, 1].  Returns (l1_loss, free_space_loss, sdf_loss) without the model's multipliers.  On the device: ONE native
    operator each way; CPU tensors (host-side checks against the reference's class) take the statement below."""
    if pred_sdf.is_cuda:
        out = _SensorDepthLoss.apply(depth_pred, depth_gt, pred_sdf, starts, directions_norm, truncation)
        return out[0], out[1], out[2]
    depth_gt = depth_gt.reshape(-1, 1)
    valid = depth_gt > 0.0
    l1 = torch.sum(valid * torch.abs(depth_gt - depth_pred.reshape(-1, 1))) / (valid.sum() + 1e-6)
    z = starts if directions_norm is None else starts / directions_norm.reshape(-1, 1)
    front = valid & (z < (depth_gt - truncation))
    back = valid & (z > (depth_gt + truncation))
    near = valid & (~front) & (~back)
    n_front, n_near = front.sum(), near.sum()
    n = n_front + n_near + 1e-6
    fs = torch.mean((torch.relu(truncation - pred_sdf) * front) ** 2) * (1.0 - n_front / n)
    sd = torch.mean(((z + pred_sdf) - depth_gt) ** 2 * near) * (1.0 - n_near / n)
    return l1, fs, sd


def _s3im_window(kernel_size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    """S3IM.create_kernel (model_components/losses.py:701-709): the outer product of the reference's (off-centre, x - size // 2) Gaussian
    of sigma 1.5, one copy per channel (grouped convolution)."""
    g = torch.tensor([math.exp(-((x - kernel_size // 2) ** 2) / float(2 * 1.5 ** 2)) for x in range(kernel_size)])
    g = (g / g.sum()).unsqueeze(1)
    w = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w.expand(channel, 1, kernel_size, kernel_size).contiguous().to(device=like.device, dtype=like.dtype)


def s3im_loss(src_vec: torch.Tensor, tar_vec: torch.Tensor, kernel_size: int = 4, stride: int = 4, repeat_time: int = 10,
              patch_height: int = 32) -> torch.Tensor:
    """S3IM (model_components/losses.py:689-771; base_surface_model.py:408-409 calls it as s3im_loss(image, outputs["rgb"])): the batch of
    N pixel colours, once in order and repeat_time - 1 times shuffled, laid out as ONE virtual image of patch_height rows; 1 - the mean
    SSIM of the two images under a kernel_size window with the given stride.  The permutations come from torch.randperm on the host's
    default generator, as in the reference (same seed, same patches).  Plain torch operators on the device (two grouped convolutions'
    worth of 30 K pixels: not a kernel of this path); N * repeat_time must be a multiple of patch_height."""
    n = len(tar_vec)
    idx = torch.cat([torch.arange(n) if i == 0 else torch.randperm(n) for i in range(repeat_time)]).to(tar_vec.device)
    tar = tar_vec[idx].permute(1, 0).reshape(1, 3, patch_height, -1)
    src = src_vec[idx].permute(1, 0).reshape(1, 3, patch_height, -1)
    w = _s3im_window(kernel_size, 3, src)
    pad = (kernel_size - 1) // 2
    conv = lambda x: torch.nn.functional.conv2d(x, w, padding=pad, groups=3, stride=stride)  # noqa: E731
    mu1, mu2 = conv(src), conv(tar)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(src * src) - mu1_sq
    sigma2_sq = conv(tar * tar) - mu2_sq
    sigma12 = conv(src * tar) - mu1_mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + c1) * (2 * sigma12 + c2)) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2))
    return 1 - ssim_map.mean()


# ---- the reference's loss OBJECTS (model_components/losses.py), for host code that constructs them as SurfaceModel.populate_modules does
# (base_surface_model.py:224-231); each is the operator above behind the reference's constructor and call signature.
class ScaleAndShiftInvariantLoss(torch.nn.Module):
    """losses.py:392-413.  forward(prediction, target, mask) on [B, H, W] images; reduction "batch-based" (the only one the path uses,
    base_surface_model.py:227)."""

    def __init__(self, alpha=0.5, scales=4, reduction="batch-based"):
        super().__init__()
        if reduction != "batch-based":
            raise NotImplementedError("ScaleAndShiftInvariantLoss: only the batch-based reduction is built")
        self.alpha, self.scales = alpha, scales

    def forward(self, prediction, target, mask):
        return scale_and_shift_invariant_loss(prediction, target, mask, self.alpha, self.scales)


class SensorDepthLoss(torch.nn.Module):
    """losses.py:628-676.  forward(batch, outputs) -> (l1_loss, free_space_loss, sdf_loss) from batch["sensor_depth"] and the surface
    model's outputs (depth, ray_samples, field_outputs[SDF], directions_norm)."""

    def __init__(self, truncation: float):
        super().__init__()
        self.truncation = truncation

    def forward(self, batch, outputs):
        from sdfstudio_amd.fields.field_heads import FieldHeadNames

        rs = outputs["ray_samples"]
        starts = rs.flat_starts if getattr(rs, "flat_starts", None) is not None else rs.frustums.starts[..., 0]
        depth = outputs["depth"]
        return sensor_depth_loss(depth, batch["sensor_depth"].to(depth.device), outputs["field_outputs"][FieldHeadNames.SDF][..., 0], starts,
                                 outputs["directions_norm"], self.truncation)


class S3IM(torch.nn.Module):
    """losses.py:689-771.  forward(src_vec, tar_vec) on [N, 3] colours."""

    def __init__(self, s3im_kernel_size=4, s3im_stride=4, s3im_repeat_time=10, s3im_patch_height=64, size_average=True):
        super().__init__()
        if not size_average:
            raise NotImplementedError("S3IM: size_average=False is not built")
        self.s3im_kernel_size, self.s3im_stride = s3im_kernel_size, s3im_stride
        self.s3im_repeat_time, self.s3im_patch_height = s3im_repeat_time, s3im_patch_height

    def forward(self, src_vec, tar_vec):
        return s3im_loss(src_vec, tar_vec, self.s3im_kernel_size, self.s3im_stride, self.s3im_repeat_time, self.s3im_patch_height)


def monosdf_normal_loss(normal_pred: torch.Tensor, normal_gt: torch.Tensor) -> torch.Tensor:
    """losses.py:264-275: L1 + cosine between the rendered and the monocular normal.  (The surface models take it inside the fused loss
    operator, surface_losses(normal_pred=..., normal_gt=...); this is the stand-alone statement for host code that calls it by name.)"""
    n_gt = torch.nn.functional.normalize(normal_gt, p=2, dim=-1)
    n_pr = torch.nn.functional.normalize(normal_pred, p=2, dim=-1)
    return torch.abs(n_pr - n_gt).sum(dim=-1).mean() + (1.0 - torch.sum(n_pr * n_gt, dim=-1)).mean()


# ---- the mip-NeRF-360 proposal loss (the BakedSDF / BakedAngelo models; NeuS-facto uses the Zip-NeRF form above)
def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    """losses.py:36-67: for every interval [t0_start, t0_end) the mass of the step function (t1, y1) over the t1 intervals it touches -
    an upper bound of the mass inside it - from the cumulative sums at the two enclosing t1 edges (two searchsorted's and two gathers)."""
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    return torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)


def lossfun_outer(t, w, t_env, w_env):
    """losses.py:70-87: (max(w - w_outer, 0))^2 / (w + 1e-7): the field's histogram (t, w) may not exceed the envelope a proposal level's
    histogram (t_env, w_env) gives it."""
    w_outer = _outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + 1e-7)


def interlevel_loss(weights_list: List[torch.Tensor], bins_list: List[torch.Tensor]) -> torch.Tensor:
    """losses.py:98-113 on the same arguments as interlevel_loss_zip: weights_list[i] [N, S_i] (last = the field's, detached here),
    bins_list[i] [N, S_i + 1] spacing-domain bin edges (ray_samples_to_sdist).  Plain torch operators on [N, S] tensors (per level: a
    cumsum, two searchsorted's, two gathers): the proposal levels' weights receive the gradient."""
    c, w = bins_list[-1].detach(), weights_list[-1].detach()
    loss = w.new_zeros(())
    for cp, wp in zip(bins_list[:-1], weights_list[:-1]):
        loss = loss + torch.mean(lossfun_outer(c, w, cp, wp))
    return loss

