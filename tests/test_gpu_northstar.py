"""north_star parity bars at BASELINE shape (-m gpu).

BASELINE.json: "outputs match the reference CPU/PyTorch path on identical rays within 1e-4 rel on rendered RGB/depth and 1e-5 on
SDF values".  tests/test_gpu_parity.py::test_field_and_render_on_reference_samples holds those bars on the reference's own
samples for the small golden network; the tests here hold them at the FULL network and sampling shape of BASELINE configs 2, 4
(NeuS-facto: 16 x 2 x 2^19 smoothstep grid, 8 x 256 + 4 x 256 MLPs, 128 samples per ray) and 1 (VolSDF, pure MLP, 64 + 32 samples),
on IDENTICAL rays and samples: the oracle runs its own sampler, and the HIP field + compositing kernels are evaluated on the
oracle's starts / ends (the end-to-end tests of test_gpu_parity.py let both sides sample for themselves and therefore carry the
drift of three fp32 resamplings).  The last test pushes the real 4096 x 128 batch of the benchmark through the kernels and compares a
64-ray subset (first, last and scattered rows) with the oracle: tail, stride and 64-bit index bugs cannot hide in small batches.

Reference lines: fields/sdf_field.py:380-410,476-525,614-689, model_components/renderers.py:81-92,245-259, cameras/rays.py:146-208.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close, product_model_from_params
from oracle import sdf_path as O
from test_gpu_parity import _bundle, _full_shape_params

pytestmark = pytest.mark.gpu

SDF_ATOL = 1e-5      # north_star: 1e-5 on SDF values
RENDER_RTOL = 1e-4   # north_star: 1e-4 relative on rendered RGB / depth (relative to the largest rendered value, plus the
#                      element-wise gate of helpers.report: 1e-3 relative on every element above 1 % of the maximum)


def _neus_facto_case(config):
    inside = config == 4
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.8 if inside else 0.5, inside_outside=inside, beta_init=0.3), num_neus_samples=128,
                     near=0.05 if inside else 0.5, far=4.0 if inside else 4.5)
    p = _full_shape_params(cfg)
    n = 64
    gen = torch.Generator().manual_seed(23)
    if inside:
        o = (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
        d = F.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        cam = torch.randint(0, 49, (n,), generator=gen)
    else:
        o, d, cam = O.synthetic_rays(n, seed=12)
    rand = [torch.rand(n, 1, generator=gen) for _ in range(3)]
    return cfg, p, o, d, cam, rand


@pytest.mark.parametrize("config", [2, 4])
def test_northstar_bars_full_shape_neus_facto_on_oracle_samples(device, config):
    from sdfstudio_amd.model_components.renderers import neus_render

    cfg, p, o, d, cam, rand = _neus_facto_case(config)
    cos_anneal = 0.4
    with torch.no_grad():
        ref = O.neus_facto_forward(o, d, cam, p, cfg, anneal=0.8, cos_anneal_ratio=cos_anneal, rand=rand, training=True)
    assert ref["starts"].shape == (64, 128)
    model = product_model_from_params(p, cfg, device).train()
    model.field.set_cos_anneal_ratio(cos_anneal)
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
    with torch.no_grad():
        sdf, grad, rgb, _ = model.field.forward_fused(rs)
        out_rgb, depth, normal, acc, weights, alpha = neus_render(
            sdf, grad, rgb, model.field.deviation_network.variance, rs.flat_directions, rs.flat_starts, rs.flat_ends, cos_anneal, None)
    fo = ref["field"]
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=SDF_ATOL)
    assert_close("alpha", alpha, fo["alpha"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("weights", weights, ref["weights"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered rgb", out_rgb, ref["rgb"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("accumulation", acc, ref["accumulation"], rtol=RENDER_RTOL, atol=1e-6)
    hit = ref["accumulation"] > 0.05
    assert int(hit.sum()) >= 16, "the case must have rays that hit the surface"
    assert_close("rendered depth", depth[hit.to(device)], ref["depth"][hit], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered normal", normal, ref["normal"], rtol=RENDER_RTOL, atol=2e-6)


def test_northstar_bars_full_shape_volsdf_on_oracle_samples(device):
    """BASELINE config 1: VolSDF, pure-MLP field (grid features off), ErrorBoundedSampler's 64 + 32 samples, eval mode."""
    from helpers import load_params
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.model_components.renderers import volsdf_render
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig

    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.1, use_grid_feature=False), proposals=())
    gen = torch.Generator().manual_seed(8)
    p = O.init_field_params(cfg.field, seed=2)
    for k in list(p):
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=gen)
    fcfg = SDFFieldConfig(bias=0.5, inside_outside=False, use_grid_feature=False, beta_init=0.1)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = VolSDFModel(VolSDFModelConfig(sdf_field=fcfg, background_model="none"), box, num_train_data=49)
    load_params(model, p)
    model = model.to(device).eval()
    n = 96
    o, d, cam = O.synthetic_rays(n, seed=9)
    with torch.no_grad():
        ref = O.volsdf_forward(o, d, cam, p, cfg, rand=None, training=False)
    assert ref["starts"].shape == (n, 96)
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
    with torch.no_grad():
        sdf, grad, rgb, _ = model.field.forward_fused(rs)
        out_rgb, depth, normal, acc, weights, density, _ = volsdf_render(sdf, grad, rgb, model.field.laplace_density.get_beta(),
                                                                         rs.flat_starts, rs.flat_ends, None)
    fo = ref["field"]
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=SDF_ATOL)
    assert_close("density", density, fo["density"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("weights", weights, ref["weights"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered rgb", out_rgb, ref["rgb"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("accumulation", acc, ref["accumulation"], rtol=RENDER_RTOL, atol=1e-6)
    hit = ref["accumulation"] > 0.05
    assert int(hit.sum()) >= 16
    assert_close("rendered depth", depth[hit.to(device)], ref["depth"][hit], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered normal", normal, ref["normal"], rtol=RENDER_RTOL, atol=2e-6)


def test_real_benchmark_batch_subset_against_oracle(device):
    """The benchmark's real 4096 x 128 batch (524 288 points, one launch per stage) through the HIP field and compositing kernels;
    64 of its rays - the first and last rows of the batch, rows around the 32- / 128-point tile seams, scattered rows - are
    compared with the oracle on the same samples at the north_star bars."""
    from sdfstudio_amd.model_components.renderers import neus_render

    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3), num_neus_samples=128)
    p = _full_shape_params(cfg, seed=7)
    n, s = 4096, 128
    o, d, cam = O.synthetic_rays(n, seed=31)
    gen = torch.Generator().manual_seed(32)
    # samples concentrated where the unit sphere is (near 2.73 - 1 .. 2.73 + 1), sorted, with a little per-ray structure
    bins = torch.sort(1.2 + 3.0 * torch.rand(n, s + 1, generator=gen), dim=-1)[0]
    starts, ends = bins[:, :-1].contiguous(), bins[:, 1:].contiguous()
    model = product_model_from_params(p, cfg, device).train()
    cos_anneal = 0.7
    model.field.set_cos_anneal_ratio(cos_anneal)
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(starts.to(device), ends.to(device))
    with torch.no_grad():
        sdf, grad, rgb, _ = model.field.forward_fused(rs)
        out_rgb, depth, normal, acc, weights, alpha = neus_render(
            sdf, grad, rgb, model.field.deviation_network.variance, rs.flat_directions, rs.flat_starts, rs.flat_ends, cos_anneal, None)
    assert sdf.shape == (n, s) and bool(torch.isfinite(out_rgb).all())
    rows = sorted(set([0, 1, 31, 32, 33, 127, 128, 4095, 4094, 4064, 4063, 3968, 3967] +
                      torch.randint(0, n, (51,), generator=gen).tolist()))[:64]
    idx = torch.tensor(rows)
    with torch.no_grad():
        fo = O.field_outputs(o[idx], d[idx], starts[idx], (ends - starts)[idx], cam[idx], p, cfg.field, None, cos_anneal, True)
        w_ref, _ = O.weights_from_alphas(fo["alpha"])
        rgb_ref, _, normal_ref, acc_ref = O.render(w_ref, fo["rgb"], fo["normal"], starts[idx], ends[idx])
    di = idx.to(device)
    assert_close("sdf (rows of the 4096 x 128 batch)", sdf[di], fo["sdf"], rtol=0, atol=SDF_ATOL)
    assert_close("alpha", alpha[di], fo["alpha"], rtol=RENDER_RTOL, atol=1e-6)
    assert_close("weights", weights[di], w_ref, rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered rgb", out_rgb[di], rgb_ref, rtol=RENDER_RTOL, atol=1e-6)
    assert_close("accumulation", acc[di], acc_ref, rtol=RENDER_RTOL, atol=1e-6)
    assert_close("rendered normal", normal[di], normal_ref, rtol=RENDER_RTOL, atol=2e-6)


def test_config4_indoor_box_collider_step_against_oracle(device):
    """BASELINE config 4's scene type: an indoor scene (cameras INSIDE the box, inside_outside = True) whose near / far planes come
    from the AABB box collider (scene_colliders.py:47-109; the reference's indoor conversions write collider_type "box",
    scripts/datasets/process_nerfstudio_to_sdfstudio.py:103), full network shape, 64 rays, training mode: planes, samples and rendered
    outputs of the product model against the oracle fed with the planes of the reference-pinned collider mirror."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H
    from sdfstudio_amd.model_components.scene_colliders import AABBBoxCollider
    from sdfstudio_amd.models.neus_facto import SceneBox
    from test_gpu_parity import _inject_facto_draws

    cfg, p, o, d, cam, rand = _neus_facto_case(4)
    model = product_model_from_params(p, cfg, device)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    model.scene_box = SceneBox(aabb=aabb, near=0.05, far=4.0, collider_type="box")
    from sdfstudio_amd.model_components.scene_colliders import build_collider

    model.collider = build_collider(model.scene_box, model.config)
    assert isinstance(model.collider, AABBBoxCollider)
    model = model.to(device).train()
    n = o.shape[0]
    nears, fars = AABBBoxCollider(model.scene_box, near_plane=0.05).train()._intersect_with_aabb(o, d, aabb)
    assert float(fars.max()) < 3.5 and float(nears.max()) == pytest.approx(0.05)  # every camera is inside: the near plane, the far wall
    cos_anneal, anneal = 0.4, 0.8
    with torch.no_grad():
        ref = O.neus_facto_forward(o, d, cam, p, cfg, anneal=anneal, cos_anneal_ratio=cos_anneal, rand=rand, training=True, nears=nears, fars=fars)
    model.field.set_cos_anneal_ratio(cos_anneal)
    model.proposal_sampler.set_anneal(anneal)
    _inject_facto_draws(model, rand, device)
    from sdfstudio_amd.cameras.rays import RayBundle

    rb = RayBundle(origins=o.to(device), directions=d.to(device), pixel_area=torch.ones(n, 1, device=device),
                   directions_norm=torch.ones(n, 1, device=device), camera_indices=cam[:, None].to(device))  # no planes: the collider sets them
    with torch.no_grad():
        out = model(rb)
    assert_close("bins", out["ray_samples"].flat_bins, ref["bins"], rtol=0, atol=5e-5)
    assert_close("euclidean sample ends", out["ray_samples"].flat_ends, ref["ends"], rtol=0, atol=1e-4)
    assert bool((out["ray_samples"].flat_ends.cpu() <= fars[:, None] + 1e-4).all()), "samples beyond the far wall of the box"
    assert_close("sdf", out["field_outputs"][H.SDF][..., 0], ref["field"]["sdf"], rtol=0, atol=1e-4)
    assert_close("rgb", out["rgb"], ref["rgb"], rtol=1e-3, atol=2e-4)
    assert_close("accumulation", out["accumulation"][..., 0], ref["accumulation"], rtol=1e-3, atol=2e-4)


def test_weight_gradients_repeat_bit_for_bit_at_full_size(device):
    """The split-K weight-gradient kernel prefetches its operands with inline-asm loads and hand-counted waits (csrc/wgrad_kernels.h):
    a wait that is one too loose reads a register before its data has landed - on SOME waves of SOME launches.  Two identical
    backward passes over the real 4096 x 128 batch must give bit-identical MLP weight gradients (their summation order is fixed;
    the hash table's gradient goes through atomics and is compared to round-off only), and they must be finite."""
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3), num_neus_samples=128)
    p = _full_shape_params(cfg, seed=9)
    n, s = 4096, 128
    o, d, cam = O.synthetic_rays(n, seed=41)
    gen = torch.Generator().manual_seed(42)
    bins = torch.sort(1.2 + 3.0 * torch.rand(n, s + 1, generator=gen), dim=-1)[0]
    model = product_model_from_params(p, cfg, device).train()
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(bins[:, :-1].contiguous().to(device), bins[:, 1:].contiguous().to(device))
    co = [torch.randn(n, s, generator=gen).to(device), (torch.randn(n, s, 3, generator=gen) * 0.3).to(device), torch.randn(n, s, 3, generator=gen).to(device)]
    runs = []
    for _ in range(3):
        model.zero_grad()
        sdf, grad, rgb, _ = model.field.forward_fused(rs)
        ((sdf * co[0]).sum() + (grad * co[1]).sum() + (rgb * co[2]).sum()).backward()
        runs.append({k: v.grad.detach().clone() for k, v in model.field.named_parameters() if v.grad is not None})
    assert len(runs[0]) >= 40
    for k, g0 in runs[0].items():
        assert bool(torch.isfinite(g0).all()), k
        for r in runs[1:]:
            if k == "encoding.params":
                assert (r[k] - g0).abs().max().item() <= 1e-4 * g0.abs().max().item(), k
            else:
                assert torch.equal(r[k], g0), f"{k}: weight gradient differs between identical launches ({(r[k] - g0).abs().max().item():.3e})"


def test_weightnorm_theta_operator_against_torch(device):
    """_ThetaFunction (one launch: weight_v / weight_g / bias of every Linear -> the flat theta; one launch back) against
    torch._weight_norm + cat under autograd, at BASELINE config 2's network shape: theta to 1 ulp-class, every v / g / bias gradient."""
    from sdfstudio_amd.fields.sdf_field import _ThetaFunction

    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3), num_neus_samples=128)
    model = product_model_from_params(_full_shape_params(cfg, seed=2), cfg, device).train()
    fld = model.field
    lins = [getattr(fld, n) for n in fld._lin_names]
    params = [t for lin in lins for t in (lin.weight_v, lin.weight_g, lin.bias)]
    theta = _ThetaFunction.apply(fld, *params)
    ref = torch.cat([t for lin in lins for t in (torch._weight_norm(lin.weight_v, lin.weight_g, 0).reshape(-1), lin.bias)])
    assert_close("theta", theta, ref, rtol=1e-6, atol=1e-7)
    co = torch.randn_like(ref)
    g_ref = torch.autograd.grad((ref * co).sum(), params)
    g_got = torch.autograd.grad((theta * co).sum(), params)
    for (name, _), a, b in zip([(f"{n}.{k}", None) for n in fld._lin_names for k in ("weight_v", "weight_g", "bias")], g_got, g_ref):
        assert_close(f"grad {name}", a, b, rtol=1e-5, atol=1e-7)


def test_surface_losses_operator_against_torch(device):
    """The fused scalar losses (L1 colour, eikonal, curvature, MonoSDF normal; model_components/losses.py surface_losses) against
    their torch statements - the reference's formulas, base_surface_model.py:399-424, neus_facto.py:312-325, losses.py:264-275 -
    values and the gradients w.r.t. every differentiable input; ragged sizes (N not a multiple of the block, zero-length gradients)."""
    from oracle.sdf_path import monosdf_normal_loss
    from sdfstudio_amd.model_components.losses import surface_losses

    gen = torch.Generator().manual_seed(7)
    n, s, delta = 777, 13, 3.1e-3
    rgb = torch.rand(n, 3, generator=gen)
    image = torch.rand(n, 3, generator=gen)
    grad = torch.randn(n, s, 3, generator=gen)
    grad[5, 3] = 0.0  # |grad| = 0: torch's norm has subgradient 0 there
    sdf = torch.randn(n, s, 1, generator=gen) * 0.1
    taps = sdf + torch.randn(n, s, 6, generator=gen) * 1e-3
    n_pred = torch.randn(n, 3, generator=gen) * 0.7
    n_gt = torch.randn(n, 3, generator=gen)

    def torch_losses(rgb, grad, sdf, taps, n_pred):
        curvature = (taps.reshape(n, s, 3, 2).sum(dim=-1) - 2 * sdf) / (delta * delta)
        return {"rgb_loss": F.l1_loss(image.to(rgb), rgb), "eikonal_loss": ((grad.norm(2, dim=-1) - 1) ** 2).mean() * 0.1,
                "curvature_loss": curvature.abs().mean() * 5e-4 * 0.37, "normal_loss": monosdf_normal_loss(n_pred, n_gt.to(rgb)) * 0.05}

    leaves_ref = [t.clone().double().requires_grad_(True) for t in (rgb, grad, sdf, taps, n_pred)]
    ref = torch_losses(*leaves_ref)
    leaves = [t.clone().to(device).requires_grad_(True) for t in (rgb, grad, sdf, taps, n_pred)]
    got = surface_losses(leaves[0], image.to(device), eik_grad=leaves[1], eikonal_mult=0.1, sdf=leaves[2], sampled_sdf=leaves[3], delta=delta,
                         curvature_mult=5e-4 * 0.37, normal_pred=leaves[4], normal_gt=n_gt.to(device), normal_mult=0.05)
    assert set(got) == set(ref)
    w = {"rgb_loss": 1.0, "eikonal_loss": 0.7, "curvature_loss": 1.3, "normal_loss": 2.1}
    for k in ref:
        assert_close(k, got[k], ref[k].float(), rtol=2e-6, atol=1e-8)
    sum(w[k] * ref[k] for k in ref).backward()
    sum(w[k] * got[k] for k in got).backward()
    for name, a, b in zip(("rgb", "eik_grad", "sdf", "sampled_sdf", "normal"), leaves, leaves_ref):
        assert_close(f"d / d {name}", a.grad, b.grad.float(), rtol=1e-5, atol=1e-9)
    # the subset NeuS-facto config 2 uses, and eval mode
    only = surface_losses(leaves[0].detach(), image.to(device), eik_grad=leaves[1].detach(), eikonal_mult=0.1)
    assert set(only) == {"rgb_loss", "eikonal_loss"}
    assert_close("rgb_loss alone", only["rgb_loss"], ref["rgb_loss"].float(), rtol=2e-6, atol=1e-8)
    assert set(surface_losses(leaves[0].detach(), image.to(device))) == {"rgb_loss"}


@pytest.mark.parametrize("n", [4096, 32 * 37])
def test_mono_depth_and_fg_mask_losses_against_torch(device, n):
    """The MonoSDF depth prior (ScaleAndShiftInvariantLoss as base_surface_model.py:427-437 calls it) and the foreground-mask BCE
    (:415-420) as native operators against their torch statements in fp64 - scale_and_shift_invariant_loss is pinned on the reference's
    class by the CPU suite: value, and the gradient w.r.t. the rendered depth INCLUDING its path through the scale / shift fit."""
    from sdfstudio_amd.model_components.losses import fg_mask_loss, monosdf_depth_loss, scale_and_shift_invariant_loss

    gen = torch.Generator().manual_seed(n)
    depth = (1.5 + torch.rand(n, 1, generator=gen) * 2.0 + 0.3 * torch.sin(torch.arange(n)[:, None] * 0.01))
    gt = (0.02 * depth + 0.004 * torch.randn(n, 1, generator=gen)).clamp_min(1e-3)  # a monocular prior: affine in the depth + noise
    ref_in = depth.clone().double().requires_grad_(True)
    mask = torch.ones(1, 32, n // 32, dtype=torch.bool)
    ref = scale_and_shift_invariant_loss(ref_in.reshape(1, 32, -1), (gt.double() * 50 + 0.5).reshape(1, 32, -1), mask, 0.5, 1)
    (ref * 1.7).backward()
    got_in = depth.clone().to(device).requires_grad_(True)
    got = monosdf_depth_loss(got_in, gt.to(device))
    (got * 1.7).backward()
    assert_close("depth loss", got, ref.float(), rtol=2e-5, atol=1e-8)
    assert_close("d depth loss / d depth", got_in.grad, ref_in.grad.float(), rtol=2e-4, atol=1e-9, elem_rtol=float("inf"))  # |x| kinks: sign flips of near-ties
    # foreground mask: weights sums on both sides of the clip, hard and soft labels
    acc = torch.rand(n, 1, generator=gen) * 1.2 - 0.1
    acc[:4, 0] = torch.tensor([0.0, 1.0, 1e-3, 1.0 - 1e-3])
    label = (torch.rand(n, 1, generator=gen) > 0.5).float()
    label[7:11] = torch.rand(4, 1, generator=gen)
    r_in = acc.clone().double().requires_grad_(True)
    r = F.binary_cross_entropy(r_in.clip(1e-3, 1.0 - 1e-3), label.double()) * 0.01
    (r * 3.0).backward()
    g_in = acc.clone().to(device).requires_grad_(True)
    gl = fg_mask_loss(g_in, label.to(device), 0.01)
    (gl * 3.0).backward()
    assert_close("fg mask loss", gl, r.float(), rtol=1e-5, atol=1e-9)
    inside = ((acc > 1e-3 + 1e-6) & (acc < 1.0 - 1e-3 - 1e-6)) | (acc < 1e-3 - 1e-6) | (acc > 1.0 - 1e-3 + 1e-6)  # fp32 / fp64 disagree AT the clip bounds
    assert_close("d fg loss / d acc", g_in.grad[inside], r_in.grad.float()[inside], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("n,s,t", [(4096, 128, 0.015), (96, 24, 0.1), (7, 1, 0.05)])
def test_sensor_depth_loss_against_torch(device, n, s, t):
    """SensorDepthLoss (model_components/losses.py:628-676) as ONE native operator each way against its torch statement in fp64 (pinned on
    the reference's class by the CPU suite): the three values and the gradients w.r.t. the per-sample sdf and the rendered depth."""
    from test_cpu_oracle_and_abi import _sensor_depth_case
    from sdfstudio_amd.model_components.losses import sensor_depth_loss

    dp, dg, sdf, st, dn = _sensor_depth_case(n + s, n, s)
    mult = torch.tensor([1.7, 0.3, 10.0])
    r_dp, r_sdf = dp.double().requires_grad_(True), sdf.double().requires_grad_(True)
    ref = torch.stack(sensor_depth_loss(r_dp, dg.double(), r_sdf, st.double(), dn.double(), t))
    (ref * mult.double()).sum().backward()
    g_dp, g_sdf = dp.to(device).requires_grad_(True), sdf.to(device).requires_grad_(True)
    got = torch.stack(sensor_depth_loss(g_dp, dg.to(device), g_sdf, st.to(device), dn.to(device), t))
    (got * mult.to(device)).sum().backward()
    # fp32 forms z = start / norm, d - t and (z + sdf) - d: a sample within an ulp of a band edge may sit on the other side in fp64
    z64 = st.double() / dn.double()
    edge = ((z64 - (dg.double()[:, None] - t)).abs() < 1e-5) | ((z64 - (dg.double()[:, None] + t)).abs() < 1e-5)
    assert int(edge.sum()) <= max(4, n * s // 2000), int(edge.sum())
    slack = 4.0 * float(edge.sum()) / (n * s) * (t + 0.05) ** 2  # what the edge samples can move a mean by
    assert_close("sensor l1", got[0], ref[0].float(), rtol=2e-6, atol=1e-9)
    assert_close("sensor free space", got[1], ref[1].float(), rtol=2e-5, atol=1e-12 + slack)
    assert_close("sensor sdf", got[2], ref[2].float(), rtol=2e-5, atol=1e-12 + slack)
    keep = ~edge
    assert_close("d / d sdf", g_sdf.grad.cpu()[keep], r_sdf.grad.float()[keep], rtol=2e-5, atol=1e-12, elem_rtol=float("inf"))
    assert_close("d / d depth", g_dp.grad, r_dp.grad.float(), rtol=2e-6, atol=1e-12)
    # rays without a measurement get no gradient at all
    assert float(g_sdf.grad.cpu()[dg <= 0].abs().max() if (dg <= 0).any() else 0.0) == 0.0


def test_rgbd_losses_through_the_model(device):
    """base_surface_model.py:439-466 through NeuSFactoModel.get_loss_dict: the three sensor-depth losses and the sparse-SfM-point loss
    appear with the reference's keys, carry the multipliers, and their backward reaches the SDF field's parameters (hash table and MLP)."""
    from helpers import load_golden, small_oracle_cfg
    from sdfstudio_amd.cameras.rays import RayBundle
    import bench as B

    model = B.build_model(device, small=True)
    c = model.config
    c.sensor_depth_l1_loss_mult, c.sensor_depth_freespace_loss_mult, c.sensor_depth_sdf_loss_mult = 0.1, 10.0, 6000.0
    c.sparse_points_sdf_loss_mult = 1.0
    c.s3im_loss_mult = 1.0  # base_surface_model.py:408-409 (512 rays x 10 repeats = a 32-row virtual image)
    model.train()
    gen = torch.Generator(device=device)
    gen.manual_seed(3)
    centers, rot = B.synthetic_cameras(device)
    o, d, norm, cam = B.draw_rays(centers, rot, 512, gen)
    out = model(RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None]))
    batch = {"image": torch.rand(512, 3, device=device, generator=gen),
             "sensor_depth": out["depth"].detach()[:, 0] * (1.0 + 0.05 * torch.randn(512, device=device, generator=gen)),
             "sparse_sfm_points": torch.randn(300, 3, device=device, generator=gen) * 0.3}
    batch["sensor_depth"][::5] = 0.0
    base = model.get_loss_dict(out, {"image": batch["image"]})
    loss = model.get_loss_dict(out, batch)
    extra = {"sensor_l1_loss", "sensor_freespace_loss", "sensor_sdf_loss", "sparse_sfm_points_sdf_loss"}
    assert set(loss) == set(base) | extra and "s3im_loss" in base
    assert 0.0 < float(loss["s3im_loss"]) < 1.0  # 1 - SSIM of a random image against the render
    extra = extra | {"s3im_loss"}
    # against the statement on the same tensors
    from sdfstudio_amd.fields.field_heads import FieldHeadNames
    from sdfstudio_amd.model_components.losses import sensor_depth_loss
    rs = out["ray_samples"]
    want = sensor_depth_loss(out["depth"].detach().cpu().double(), batch["sensor_depth"].cpu().double(),
                             out["field_outputs"][FieldHeadNames.SDF][..., 0].detach().cpu().double(), rs.flat_starts.cpu().double(),
                             out["directions_norm"].cpu().double(), c.sensor_depth_truncation)
    for k, w, m in zip(("sensor_l1_loss", "sensor_freespace_loss", "sensor_sdf_loss"), want, (0.1, 10.0, 6000.0)):
        assert_close(k, loss[k], (w * m).float(), rtol=5e-5, atol=1e-9)
    for p in model.parameters():
        p.grad = None
    sum(loss[k] for k in extra).backward()
    # (the hash table itself has an exactly zero gradient here: the geometric initialisation zeroes layer 0's feature columns, sdf_field.py:300-303)
    for name in ("glin0", "glin4", "glin8"):
        g = getattr(model.field, name).weight_v.grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0.0, name
    # layer 0's FEATURE columns receive gradient (the features are not zero), which is what moves the table from the second step on
    assert float(model.field.glin0.weight_v.grad[:, 3:].abs().max()) > 0.0


def test_gradient_slots_give_the_same_gradients_as_plain_autograd(device):
    """grad_slots.py end to end on the golden NeuS-facto model: with a FlatGradients the native backward kernels write parameter
    gradients straight into the flat buffer (no AccumulateGrad launch); the flat buffer must equal what plain autograd leaves in
    .grad without one - bit for bit for the MLP parameters (fixed summation order), to round-off for the hash tables (atomics)."""
    from helpers import load_golden, small_oracle_cfg
    from sdfstudio_amd.distributed import FlatGradients

    g = load_golden("train")
    cfg = small_oracle_cfg()
    rb = lambda: _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)  # noqa: E731

    def step(model, flat):
        torch.manual_seed(3)
        out = model(rb())
        loss = sum(model.get_loss_dict(out, {"image": g["in"]["image"]}).values())
        if flat is None:
            model.zero_grad()
        else:
            flat.zero()
        loss.backward()
        if flat is not None:
            flat.finish()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    plain = step(product_model_from_params(g["param"], cfg, device).train(), None)
    model = product_model_from_params(g["param"], cfg, device).train()
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    flat = FlatGradients([p for grp in groups.values() for p in grp], buckets=list(groups.values()))
    for rep in range(2):  # the second pass starts from adopted views: zero() must have detached them again
        slotted = step(model, flat)
        assert set(slotted) >= set(plain)
        for k, ref in plain.items():
            p = dict(model.named_parameters())[k]
            assert p.grad.data_ptr() == flat.flat.data_ptr() + 4 * flat._offset[id(p)], f"{k}: .grad does not alias the flat buffer"
            if k.endswith("table") or k.endswith("encoding.params") or "deviation_network" in k or "laplace_density" in k:
                # accumulated with atomics (hash tables; the variance's per-ray contributions): the summation order differs from
                # launch to launch, and single entries with cancelling contributions move by ~1e-4 of the tensor's maximum
                assert (slotted[k] - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-12, k
            else:
                assert torch.equal(slotted[k], ref), f"{k}: {(slotted[k] - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_field_forward_is_bit_reproducible(device, mode):
    """The fused forward has no atomics: every call must return the SAME BITS, whatever the allocator's free blocks hold and whatever ran
    on the CUs before.  This is the test a tolerance cannot replace: until the end of round 3 one call in five of exactly this forward
    returned a workgroup's colours off by ~5e-5 - a gemm started on a weight chunk whose LDS-DMA was still landing (the counted
    s_waitcnt of mlp_core.h) - and every parity test passed."""
    from helpers import load_golden, small_oracle_cfg

    g = load_golden(mode)
    cfg = small_oracle_cfg()
    ref = None
    for it, fill in enumerate((0.0, float("nan"), 1e30, 0.0, float("nan"), -3.7)):
        blocks = [torch.full((n,), fill, device=device) for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
        del blocks
        model = product_model_from_params(g["param"], cfg, device).train(mode == "train")
        model.field.set_cos_anneal_ratio(float(g["in"]["cos_anneal"]))
        rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
        rs = rb.get_ray_samples(g["out"]["starts"].to(device), g["out"]["ends"].to(device))
        for rep in range(5):
            with torch.no_grad():
                out = [t.clone() for t in model.field.forward_fused(rs)[:3]]
            if ref is None:
                ref = out
            for name, a, b in zip(("sdf", "gradient", "rgb"), out, ref):
                assert torch.equal(a, b), f"{name}: call {it}.{rep} differs from the first call in {int((a != b).sum())} elements, max {float((a - b).abs().max()):.3e}"


def test_no_grad_forward_runs_the_nothing_saved_kernels(device):
    """Rendering under torch.no_grad() must take the forward-only path: nothing saved (no r_l stores in the chain launch) and the
    forward-only workspace.  Function.forward cannot see the caller's grad mode and ctx.needs_input_grad reports requires_grad of the
    inputs regardless of it, so until round 5 the hash table (a Parameter) made every eval render run the TRAINING kernels on the
    training workspace (kernel names in profiles/r5a_eval_kernel_stats.csv).  Same bits either way; a quarter of the memory."""
    from helpers import load_golden, small_oracle_cfg
    from sdfstudio_amd import _lib

    g = load_golden("eval")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).eval()
    model.field.set_cos_anneal_ratio(float(g["in"]["cos_anneal"]))
    n, s = 256, 64
    gen = torch.Generator().manual_seed(3)
    o = torch.randn(n, 3, generator=gen) * 0.3
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    rb = _bundle(o, d, torch.zeros(n, dtype=torch.long), cfg.near, cfg.far, device)
    starts = torch.sort(torch.rand(n, s, generator=gen) * 3.0 + 0.5, dim=-1)[0].to(device)
    rs = rb.get_ray_samples(starts, starts + 0.05)
    with_graph = [t.detach().clone() for t in model.field.forward_fused(rs)[:3]]  # parameters require grad: the saving kernels
    lib = _lib.load()
    train_ws = lib.sdfhip_field_workspace_size(model.field._handle, n * s, 1)
    infer_ws = lib.sdfhip_field_workspace_size(model.field._handle, n * s, 2)
    assert infer_ws * 2 < train_ws
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = [t.clone() for t in model.field.forward_fused(rs)[:3]]
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < infer_ws + (train_ws - infer_ws) // 4, f"no_grad forward allocated {peak} B: training workspace is {train_ws} B, forward-only {infer_ws} B"
    for name, a, b in zip(("sdf", "gradient", "rgb"), out, with_graph):
        assert torch.equal(a, b), f"{name}: no_grad forward differs from the saving forward in {int((a != b).sum())} elements"
