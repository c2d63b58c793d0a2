"""GPU parity tests (-m gpu): every HIP entry point against the CPU oracle on seeded inputs, and the whole
NeuS-facto step against the golden vectors minted from the reference's own Python.

Tolerances (north_star): 1e-5 on SDF values, 1e-4 relative on rendered RGB / depth; gradients 1e-3 relative to the
largest entry of each tensor (fp32 accumulation order differs between MFMA tiles / atomics and the CPU BLAS).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import (assert_close, assert_fp32_class, assert_grads_close_mod_relu_flips, load_golden, load_golden_file, product_grads,
                     product_model_from_params, relu_flip_basis, small_oracle_cfg, to_double)
from oracle import sdf_path as O

pytestmark = pytest.mark.gpu


def _bundle(origins, dirs, cam, near, far, device):
    from sdfstudio_amd.cameras.rays import RayBundle

    n = origins.shape[0]
    return RayBundle(
        origins=origins.to(device), directions=dirs.to(device), pixel_area=torch.ones(n, 1, device=device),
        directions_norm=torch.ones(n, 1, device=device), camera_indices=cam[:, None].to(device),
        nears=torch.full((n, 1), near, device=device), fars=torch.full((n, 1), far, device=device),
    )


# ------------------------------------------------------------------------------------------------ samplers
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("S", [32, 256])
def test_spaced_sampler(device, training, S):
    from sdfstudio_amd.model_components.ray_samplers import UniformLinDispPiecewiseSampler

    torch.manual_seed(0)
    n = 130
    o, d, cam = O.synthetic_rays(n)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    jit = torch.rand(n, 1)
    smp = UniformLinDispPiecewiseSampler(single_jitter=True).train(training)
    smp.jitter_override = jit.to(device)
    rs = smp(rb, num_samples=S)
    bins = O.initial_bins(n, S, jit if training else None)
    eu = O.spacing_to_euclidean(bins, torch.full((n,), 0.5), torch.full((n,), 4.5))
    assert_close("bins", rs.flat_bins, bins, rtol=0, atol=2e-7)
    assert_close("starts", rs.flat_starts, eu[:, :-1], rtol=1e-6, atol=1e-6)
    assert_close("ends", rs.flat_ends, eu[:, 1:], rtol=1e-6, atol=1e-6)
    assert rs.frustums.starts.shape == (n, S, 1) and rs.deltas.shape == (n, S, 1)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(256, 96), (96, 128), (32, 24), (7, 5)])
def test_pdf_sampler(device, training, shape):
    from sdfstudio_amd.model_components.ray_samplers import PDFSampler, UniformLinDispPiecewiseSampler

    s_in, s_out = shape
    torch.manual_seed(1)
    n = 67
    o, d, cam = O.synthetic_rays(n)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    rs0 = UniformLinDispPiecewiseSampler(single_jitter=True).eval()(rb, num_samples=s_in)
    w = torch.rand(n, s_in) ** 4
    w[3] = 0.0  # a ray with zero weight exercises the eps padding (ray_samplers.py:307-310)
    jit = torch.rand(n, 1)
    pdf = PDFSampler(single_jitter=True, include_original=False).train(training)
    pdf.jitter_override = jit.to(device)
    rs = pdf(rb, rs0, w.to(device)[..., None], num_samples=s_out, anneal=0.7)
    bins_in = O.initial_bins(n, s_in, None)
    ref = O.pdf_sample(torch.pow(w, 0.7), bins_in, s_out, jit if training else None)
    # Inverse-CDF sampling divides by cdf[i+1] - cdf[i] (>= 0.01 / sum(w + 0.01) ~ 2e-4 here): one fp32 ulp of the cdf
    # (1e-7; the wave scan and torch.cumsum / torch.sum add in different orders) moves a bin edge by up to ~5e-6.
    # The same spread separates torch's own CPU and GPU cumsum, so 2e-5 absolute on the [0,1] spacing bins is the bar.
    assert_close("pdf bins", rs.flat_bins, ref, rtol=0, atol=2e-5)
    b = rs.flat_bins
    assert (b[:, 1:] >= b[:, :-1]).all() and b.min() >= 0 and b.max() <= 1
    # spacing -> euclidean is checked on the kernel's own bins (tight), so the conditioning above does not enter
    eu = O.spacing_to_euclidean(b.cpu(), torch.full((n,), 0.5), torch.full((n,), 4.5))
    assert_close("pdf starts", rs.flat_starts, eu[:, :-1], rtol=2e-6, atol=2e-6)
    assert_close("pdf ends", rs.flat_ends, eu[:, 1:], rtol=2e-6, atol=2e-6)


# ------------------------------------------------------------------------------------------------ NeuS sampler
@pytest.mark.parametrize("single_jitter", [True, False])
@pytest.mark.parametrize("kind", ["piecewise", "uniform", "lindisp", "sqrt", "log"])
def test_spaced_sampler_family(device, kind, single_jitter):
    """Every SpacedSampler subclass (ray_samplers.py:130-247), one draw per ray and one per bin edge (:105-113), train + eval."""
    from sdfstudio_amd.model_components import ray_samplers as RS

    cls = {"piecewise": RS.UniformLinDispPiecewiseSampler, "uniform": RS.UniformSampler, "lindisp": RS.LinearDisparitySampler,
           "sqrt": RS.SqrtSampler, "log": RS.LogSampler}[kind]
    torch.manual_seed(3)
    n, S = 67, 48
    o, d, cam = O.synthetic_rays(n)
    nears, fars = 0.05 + torch.rand(n), 2.0 + 100.0 * torch.rand(n)  # the background sampler runs to far = 1000 (base_surface_model.py)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    rb.nears, rb.fars = nears[:, None].to(device), fars[:, None].to(device)
    for training in (True, False):
        jit = torch.rand(n, 1) if single_jitter else torch.rand(n, S + 1)
        smp = cls(single_jitter=single_jitter).train(training)
        smp.jitter_override = jit.to(device)
        rs = smp(rb, num_samples=S)
        bins = O.initial_bins(n, S, jit if training else None)
        eu = O.spaced_to_euclidean(kind, bins, nears, fars)
        assert_close(f"{kind} bins", rs.flat_bins, bins, rtol=0, atol=2e-7)
        assert_close(f"{kind} starts", rs.flat_starts, eu[:, :-1], rtol=2e-6, atol=1e-6)
        assert_close(f"{kind} ends", rs.flat_ends, eu[:, 1:], rtol=2e-6, atol=1e-6)
        x = torch.rand(n, 5)  # applied to [N, bins] tensors, as the PDF sampler does (ray_samplers.py:359)
        assert_close(f"{kind} spacing_to_euclidean_fn", rs.spacing_to_euclidean_fn(x.to(device)),
                     O.spaced_to_euclidean(kind, x, nears, fars), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("include_original", [False, True])
@pytest.mark.parametrize("single_jitter", [True, False])
@pytest.mark.parametrize("kind", ["piecewise", "lindisp"])
def test_pdf_sampler_general(device, kind, single_jitter, include_original):
    """PDFSampler in a non-default spacing domain, per-bin-edge jitter (ray_samplers.py:321-326) and include_original=True (:354-355)
    against the oracle (pinned on the reference's PDFSampler by test_pdf_sampler_oracle_against_reference)."""
    from sdfstudio_amd.model_components import ray_samplers as RS

    torch.manual_seed(9)
    n, s_in, s_out = 41, 48, 29
    o, d, cam = O.synthetic_rays(n)
    nears, fars = 0.5 + torch.rand(n), 3.0 + 20 * torch.rand(n)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    rb.nears, rb.fars = nears[:, None].to(device), fars[:, None].to(device)
    base = (RS.UniformLinDispPiecewiseSampler if kind == "piecewise" else RS.LinearDisparitySampler)(single_jitter=True).eval()(rb, num_samples=s_in)
    w = torch.rand(n, s_in) * (torch.rand(n, s_in) < 0.6)
    w[2] = 0.0
    for training in (True, False):
        jit = torch.rand(n, 1) if single_jitter else torch.rand(n, s_out + 1)
        smp = RS.PDFSampler(num_samples=s_out, single_jitter=single_jitter, include_original=include_original, spacing=kind).train(training)
        smp.jitter_override = jit.to(device)
        rs = smp(rb, base, w[..., None].to(device))
        existing = base.flat_bins.cpu()
        bins = O.pdf_sample(w, existing, s_out, jit if training else None)
        if include_original:
            bins = torch.sort(torch.cat([existing, bins], -1), -1)[0]
        eu = O.spaced_to_euclidean(kind, bins, nears, fars)
        # one fp32 ulp of the cdf (a cumsum of ~50 terms) moves a bin edge by ~5e-6 of the spacing range
        assert_close("bins", rs.flat_bins, bins, rtol=0, atol=1e-5)
        assert_close("starts", rs.flat_starts, eu[:, :-1], rtol=1e-4, atol=1e-5)
        assert_close("ends", rs.flat_ends, eu[:, 1:], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("training", [True, False])
def test_proposal_sampler_with_uniform_initial_sampler(device, training):
    """ProposalNetworkSampler(use_uniform_sampler=True) (ray_samplers.py:517-522): UniformSampler bins, then PDF resampling in the SAME
    (uniform) spacing domain, twice; against the oracle's statements composed the same way, with analytic density functions."""
    from sdfstudio_amd.model_components.ray_samplers import ProposalNetworkSampler

    n, counts, s_final = 37, (48, 24), 16
    o, d, cam = O.synthetic_rays(n, seed=4)
    near, far = 0.5, 4.5
    rb = _bundle(o, d, cam, near, far, device)
    peaks = [1.7, 2.4]
    smp = ProposalNetworkSampler(num_proposal_samples_per_ray=counts, num_nerf_samples_per_ray=s_final, num_proposal_network_iterations=2,
                                 use_uniform_sampler=True, single_jitter=True).train(training)
    gen = torch.Generator().manual_seed(17)
    t0, u = torch.rand(n, 1, generator=gen), torch.rand(n, 1, generator=gen)
    smp.initial_sampler.jitter_override = t0.to(device)
    smp.pdf_sampler.jitter_override = u.to(device)
    fns = [lambda rs, c=c: 6.0 * torch.exp(-4.0 * ((rs.frustums.starts + rs.frustums.ends) / 2 - c) ** 2) for c in peaks]
    rs, weights_list, samples_list = smp(rb, density_fns=fns)
    nears, fars = torch.full((n,), near), torch.full((n,), far)
    bins = O.initial_bins(n, counts[0], t0 if training else None)
    for lvl, c in enumerate(peaks):
        eu = O.uniform_to_euclidean(bins, nears, fars)
        assert_close(f"level {lvl} bins", samples_list[lvl].flat_bins, bins, rtol=0, atol=1e-5)
        dens = 6.0 * torch.exp(-4.0 * ((eu[:, :-1] + eu[:, 1:]) / 2 - c) ** 2)
        w = O.weights_from_density(dens, eu[:, 1:] - eu[:, :-1])
        assert_close(f"level {lvl} weights", weights_list[lvl][..., 0], w, rtol=1e-4, atol=1e-6)
        bins = O.pdf_sample(w, bins, counts[1] if lvl == 0 else s_final, u if training else None)
    eu = O.uniform_to_euclidean(bins, nears, fars)
    assert_close("final bins", rs.flat_bins, bins, rtol=0, atol=2e-5)
    assert_close("final starts", rs.flat_starts, eu[:, :-1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("training", [True, False])
def test_uniform_sampler(device, training):
    from sdfstudio_amd.model_components.ray_samplers import UniformSampler

    n, S = 77, 64
    o, d, cam = O.synthetic_rays(n)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    jit = torch.rand(n, 1)
    smp = UniformSampler(single_jitter=True).train(training)
    smp.jitter_override = jit.to(device)
    rs = smp(rb, num_samples=S)
    bins = O.initial_bins(n, S, jit if training else None)
    eu = O.uniform_to_euclidean(bins, torch.full((n,), 0.5), torch.full((n,), 4.5))
    assert_close("bins", rs.flat_bins, bins, rtol=0, atol=2e-7)
    assert_close("starts", rs.flat_starts, eu[:, :-1], rtol=1e-6, atol=1e-6)
    assert_close("ends", rs.flat_ends, eu[:, 1:], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_neus_upsample_steps_against_reference(device, mode):
    """Every up-sampling step of the reference's NeuSSampler run (golden: its own inputs and outputs per step): merged sdf,
    new bins (PDF resampling), merged bins and the merge index, on identical inputs."""
    from sdfstudio_amd.model_components.ray_samplers import NeuSSampler

    g = load_golden_file(f"neus_small_{mode}.npz")
    cfg = small_oracle_cfg()
    n = g["in"]["origins"].shape[0]
    steps, n_imp, base = int(g["in"]["steps"]), int(g["in"]["num_importance"]), float(g["in"]["base_variance"])
    smp = NeuSSampler(num_samples=int(g["in"]["num_samples"]), num_samples_importance=n_imp, num_upsample_steps=steps,
                      base_variance=base).train(mode == "train")
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    for it in range(steps):
        st = g[f"step{it}"]
        bins_in, sdf_in = st["bins_in"].to(device), st["sdf_in"].to(device)
        jit = g["in"][f"rand{1 + it}"].to(device).reshape(-1) if mode == "train" else None
        sdf_m, new_bins, new_starts, new_ends, m_bins, m_index, m_starts, m_ends = smp.upsample_step(
            rb, bins_in, sdf_in, None, None, n_imp // steps, base * 2 ** it, jit)
        assert_close(f"step {it} merged sdf (identity)", sdf_m, st["sdf_in"], rtol=0, atol=0)
        # inverse-CDF with histogram_padding 1e-5: cdf increments down to ~1e-6, one fp32 ulp of the cdf moves an edge by 1e-4 of a bin
        assert_close(f"step {it} new bins", new_bins, st["new_bins"], rtol=0, atol=3e-5)
        ref_m, ref_i = O.merge_bins(bins_in.cpu(), new_bins.cpu())  # merge is exact given the kernel's own new bins
        assert_close(f"step {it} merged bins", m_bins, ref_m, rtol=0, atol=0)
        assert torch.equal(m_index.cpu().long(), ref_i), f"step {it}: merge index"
        eu = O.uniform_to_euclidean(ref_m, torch.full((n,), cfg.near), torch.full((n,), cfg.far))
        assert_close(f"step {it} merged starts", m_starts, eu[:, :-1], rtol=1e-6, atol=1e-6)
        assert_close(f"step {it} merged ends", m_ends, eu[:, 1:], rtol=1e-6, atol=1e-6)
        # and the sdf gather with a real index: feed (sdf_a, sdf_b, index) of the NEXT reference step
        if it + 1 < steps:
            nxt = g[f"step{it + 1}"]
            idx = st["index"].to(device).int().contiguous()
            cat_ref = torch.gather(torch.cat([st["sdf_in"], torch.zeros(n, n_imp // steps)], -1), 1, st["index"])
            known = st["index"] < st["sdf_in"].shape[1]
            sdf_b = torch.zeros(n, n_imp // steps)
            # sdf of the new samples = the entries of the next step's merged sdf that came from list 2
            sdf_b[torch.arange(n)[:, None].expand_as(st["index"])[~known], (st["index"] - st["sdf_in"].shape[1])[~known]] = \
                nxt["sdf_in"][~known]
            out = smp.upsample_step(rb, nxt["bins_in"].to(device), sdf_in, sdf_b.to(device), idx, n_imp // steps,
                                    base * 2 ** (it + 1), None if jit is None else g["in"][f"rand{2 + it}"].to(device).reshape(-1))
            assert_close(f"step {it + 1} merged sdf (gather)", out[0], nxt["sdf_in"], rtol=0, atol=0)
            assert cat_ref.shape == nxt["sdf_in"].shape


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_neus_model_against_reference_golden(device, mode):
    """models/neus.py end to end.  Sampler: four rounds of ill-conditioned inverse-CDF resampling -> the bulk of the samples
    must agree to fp32 round-off, none may be off by more than a few 1e-3 (the oracle's own spread against the reference);
    field + renderer + losses + gradients: on the reference's samples, 1e-5 on SDF / 1e-4 relative on the rest."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.model_components.renderers import neus_render
    from sdfstudio_amd.models.neus import NeuSModel, NeuSModelConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from helpers import load_params

    g = load_golden_file(f"neus_small_{mode}.npz")
    cfg = small_oracle_cfg()
    fc = cfg.field
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
                          num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
                          inside_outside=fc.inside_outside, use_grid_feature=True, beta_init=fc.beta_init, num_levels=fc.num_levels,
                          max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
                          hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep)
    steps = int(g["in"]["steps"])
    mcfg = NeuSModelConfig(sdf_field=fcfg, num_samples=int(g["in"]["num_samples"]),
                           num_samples_importance=int(g["in"]["num_importance"]), num_up_sample_steps=steps,
                           base_variance=float(g["in"]["base_variance"]), eikonal_loss_mult=cfg.eikonal_loss_mult, background_model="none")
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = NeuSModel(mcfg, box, num_train_data=49)
    load_params(model, g["param"])
    training = mode == "train"
    model = model.to(device).train(training)
    ca = float(g["in"]["cos_anneal"])
    model.field.set_cos_anneal_ratio(ca)
    model.sampler.uniform_sampler.jitter_override = g["in"]["rand0"].to(device)
    model.sampler.jitter_overrides = [g["in"][f"rand{1 + i}"].to(device) for i in range(steps)]
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    ref = g["out"]
    out = model(rb)
    d_bins = (out["ray_samples"].flat_bins.cpu() - ref["bins"]).abs() if training else None
    if training:
        assert d_bins.median().item() <= 1e-5 and d_bins.max().item() <= 5e-3, (d_bins.median().item(), d_bins.max().item())
        b = out["ray_samples"].flat_bins
        assert (b[:, 1:] >= b[:, :-1]).all() and b.min() >= 0 and b.max() <= 1
    assert_close("rgb (own samples)", out["rgb"], ref["rgb"], rtol=5e-3, atol=5e-3)
    # identical samples
    rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
    sdf, grad, rgb, x = model.field.forward_fused(rs)
    out_rgb, depth, normal, acc, weights, alpha = neus_render(
        sdf, grad, rgb, model.field.deviation_network.variance, rs.flat_directions, rs.flat_starts, rs.flat_ends, ca, None)
    assert_close("sdf", sdf, ref["sdf"], rtol=0, atol=1e-5)
    assert_close("gradient", grad, ref["gradient"], rtol=1e-4, atol=1e-6)
    assert_close("alpha", alpha, ref["alpha"], rtol=1e-4, atol=1e-6)
    assert_close("weights", weights, ref["weights"], rtol=1e-4, atol=1e-6)
    assert_close("rendered rgb", out_rgb.clamp(0, 1) if not training else out_rgb, ref["rgb"], rtol=1e-4, atol=1e-6)
    hit = ref["accumulation"] > 0.05
    assert_close("rendered depth", depth[hit.to(device)], ref["depth"][hit], rtol=1e-4, atol=1e-6)
    assert_close("rendered normal", normal, ref["normal"], rtol=1e-4, atol=1e-6)
    if training:
        loss = F.l1_loss(g["in"]["image"].to(device), out_rgb) + ((grad.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
        assert_close("loss", loss, g["loss"]["rgb_loss"] + g["loss"]["eikonal_loss"], rtol=1e-4, atol=1e-7)
        model.zero_grad()
        loss.backward()
        got = product_grads(model)
        for k in g["grad"]:
            assert k in got, f"no gradient for {k}"
        assert len(g["grad"]) >= 40

        # The colour network's ReLUs: pre-activations within round-off of zero may take either branch (helpers.py).  Its raw
        # d sdf / dx input carries ~1e-4 of fp32 noise here (finest hash levels: scale 2e3 x one ulp of the position; the
        # "gradient" line above), which reaches the pre-activations as ~1e-5: that is the knife-edge margin.
        def oracle_backward():
            po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
            o = O.neus_forward(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], po, cfg, cos_anneal_ratio=ca, training=True,
                               samples=(ref["bins"], ref["starts"], ref["ends"]))
            lo = F.l1_loss(o["rgb"], g["in"]["image"]) + ((o["field"]["gradient"].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
            lo.backward()
            return {k: po[k].grad for k in g["grad"]}

        _, basis = relu_flip_basis(oracle_backward, margin=3e-5)
        assert_grads_close_mod_relu_flips(got, g["grad"], basis, rtol=5e-3)


def test_neus_sampler_per_sample_jitter(device):
    """NeuSSampler(single_jitter=False) (ray_samplers.py:825, 836-840): the initial UniformSampler draws one offset per bin edge
    (:107-110) and every PDF resampling one per NEW bin edge (:321-330).  Whole sampler against the oracle's, on a sphere sdf
    that both sides evaluate at the frustum start positions."""
    from sdfstudio_amd.model_components.ray_samplers import NeuSSampler

    n, S, n_imp, steps = 41, 24, 32, 4
    o, d, cam = O.synthetic_rays(n, seed=8)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    gen = torch.Generator().manual_seed(3)
    rand = [torch.rand(n, S + 1, generator=gen)] + [torch.rand(n, n_imp // steps + 1, generator=gen) for _ in range(steps)]
    smp = NeuSSampler(num_samples=S, num_samples_importance=n_imp, num_upsample_steps=steps, single_jitter=False).train(True)
    smp.uniform_sampler.jitter_override = rand[0].to(device)
    smp.jitter_overrides = [r.to(device) for r in rand[1:]]
    rs = smp(rb, sdf_fn=lambda r: r.frustums.get_start_positions().norm(dim=-1, keepdim=True) - 1.0)
    nears, fars = torch.full((n,), 0.5), torch.full((n,), 4.5)
    bins, starts, ends = O.neus_sampler(o, d, nears, fars, lambda t: (o[:, None, :] + d[:, None, :] * t[..., None]).norm(dim=-1) - 1.0,
                                        num_samples=S, num_samples_importance=n_imp, num_upsample_steps=steps, rand=rand)
    # histogram_padding 1e-5: cdf increments down to ~1e-6, one fp32 ulp of the cdf moves an edge by 1e-4 of a bin (see the step test)
    assert_close("bins", rs.flat_bins, bins, rtol=0, atol=1e-4)
    assert_close("starts", rs.flat_starts, starts, rtol=0, atol=5e-4)
    assert float((rs.flat_bins[:, 1:] - rs.flat_bins[:, :-1]).min()) >= 0.0
    # and the draws really are per sample: a single-jitter run with the first column of every draw gives different bins
    smp1 = NeuSSampler(num_samples=S, num_samples_importance=n_imp, num_upsample_steps=steps, single_jitter=True).train(True)
    smp1.uniform_sampler.jitter_override = rand[0][:, :1].to(device)
    smp1.jitter_overrides = [r[:, 0].to(device) for r in rand[1:]]
    rs1 = smp1(rb, sdf_fn=lambda r: r.frustums.get_start_positions().norm(dim=-1, keepdim=True) - 1.0)
    assert float((rs1.flat_bins - rs.flat_bins).abs().max()) > 1e-3


# ------------------------------------------------------------------------------------------------ VolSDF sampler
def test_uniform_sampler_per_sample_jitter(device):
    from sdfstudio_amd.model_components.ray_samplers import UniformSampler

    n, S = 33, 40
    o, d, cam = O.synthetic_rays(n)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    jit = torch.rand(n, S + 1)
    smp = UniformSampler(single_jitter=False).train(True)
    smp.jitter_override = jit.to(device)
    rs = smp(rb, num_samples=S)
    bins = O.initial_bins(n, S, jit)
    assert_close("bins", rs.flat_bins, bins, rtol=0, atol=2e-7)
    eu = O.uniform_to_euclidean(bins, torch.full((n,), 0.5), torch.full((n,), 4.5))
    assert_close("starts", rs.flat_starts, eu[:, :-1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_volsdf_sampler_steps_against_reference(device, mode):
    """Every outer iteration of the reference's ErrorBoundedSampler run (golden: its own inputs / outputs per iteration):
    beta after the bisection, weights, error-proportional weights, PDF resampling, merge -- on identical inputs."""
    from sdfstudio_amd.model_components.ray_samplers import ErrorBoundedSampler

    g = load_golden_file(f"volsdf_small_{mode}.npz")
    cfg = small_oracle_cfg()
    n = g["in"]["origins"].shape[0]
    n_iters = int(g["in"]["n_iters"])
    ns, ns_eval, ns_extra = int(g["in"]["num_samples"]), int(g["in"]["num_samples_eval"]), int(g["in"]["num_samples_extra"])
    smp = ErrorBoundedSampler(num_samples=ns, num_samples_eval=ns_eval, num_samples_extra=ns_extra).train(mode == "train")
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    beta0 = (g["param"]["laplace_density.beta"].abs() + g["param"]["laplace_density.beta_min"]).to(device)
    rand = [g["in"][f"rand{i}"] for i in range(len([k for k in g["in"] if k.startswith("rand")]))]
    assert n_iters >= 3
    for it in range(n_iters):
        st = g[f"step{it}"]
        bins_in = st["bins_in"].to(device)
        sdf_m, beta, weights, err_w, flag = smp.bound_step(rb, bins_in, st["sdf_in"].to(device), None, None,
                                                           st["beta_in"].to(device), beta0)
        assert_close(f"it {it} beta", beta, st["beta_out"], rtol=2e-5, atol=0)
        assert_close(f"it {it} weights", weights, st["weights"], rtol=0, atol=5e-6)
        assert bool(flag.item()) == bool((st["beta_out"] > beta0.cpu()).any())
        if "err_weights" in st:
            assert_close(f"it {it} error weights", err_w, st["err_weights"], rtol=2e-4, atol=1e-6)
            smp.jitter_queue = [rand[1 + it]] if mode == "train" else None
            new_bins, new_starts, _ = smp._pdf(rb, st["err_weights"].to(device), bins_in, ns_eval)
            # error weights span ~10 orders of magnitude: cdf increments of ~1e-7 in the tails (see test_pdf_sampler)
            d = (new_bins.cpu() - st["new_bins"]).abs()
            assert d.median().item() <= 1e-6 and d.max().item() <= 2e-3, (d.median().item(), d.max().item())
            m_bins, m_index, m_starts, m_ends = smp.merge(rb, bins_in, st["new_bins"].to(device))
            assert_close(f"it {it} merged bins", m_bins, st["merged_bins"], rtol=0, atol=0)
            # the index must be a permutation that reproduces the merged starts; on exact ties (deterministic eval-mode bins
            # coincide with resampled edges) torch.sort's order is unspecified, so equality with the reference index is only
            # required where the merged value is unique
            mi = m_index.cpu().long()
            cat = torch.cat([st["bins_in"][:, :-1], st["new_bins"][:, :-1]], -1)
            assert torch.equal(torch.gather(cat, 1, mi), st["merged_bins"][:, :-1]), f"it {it}: merge index does not reproduce the bins"
            assert torch.equal(torch.sort(mi, -1)[0], torch.arange(mi.shape[1])[None].expand_as(mi)), f"it {it}: not a permutation"
            mbv = st["merged_bins"][:, :-1]
            uniq = torch.ones_like(mbv, dtype=torch.bool)
            uniq[:, 1:] &= mbv[:, 1:] != mbv[:, :-1]
            uniq[:, :-1] &= mbv[:, 1:] != mbv[:, :-1]
            assert torch.equal(mi[uniq], st["index"][uniq]), f"it {it}: merge index"
        else:
            smp.jitter_queue = [rand[1 + it]] if mode == "train" else None
            f_bins, _, _ = smp._pdf(rb, st["weights"].to(device), bins_in, ns)
            assert_close(f"it {it} final bins", f_bins, st["final_bins"], rtol=0, atol=3e-5)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_volsdf_model_against_reference_golden(device, mode):
    """BASELINE config 1 flavour (VolSDF, pure-MLP field: zero grid features) end to end; the golden comes from the reference
    with no tiny-cuda-nn shim in the loop.  Field + density rendering + gradients on the reference's samples; the sampler end to
    end within its fp32 conditioning."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig
    from sdfstudio_amd.model_components.renderers import density_to_weights
    from helpers import load_params

    g = load_golden_file(f"volsdf_small_{mode}.npz")
    cfg = small_oracle_cfg()
    fc = cfg.field
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
                          num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
                          inside_outside=fc.inside_outside, use_grid_feature=False, beta_init=fc.beta_init, num_levels=fc.num_levels,
                          max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
                          hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep)
    mcfg = VolSDFModelConfig(sdf_field=fcfg, num_samples=int(g["in"]["num_samples"]), num_samples_eval=int(g["in"]["num_samples_eval"]),
                             num_samples_extra=int(g["in"]["num_samples_extra"]), eikonal_loss_mult=cfg.eikonal_loss_mult, background_model="none")
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = VolSDFModel(mcfg, box, num_train_data=49)
    load_params(model, g["param"])
    training = mode == "train"
    model = model.to(device).train(training)
    rand = [g["in"][f"rand{i}"] for i in range(len([k for k in g["in"] if k.startswith("rand")]))]
    model.sampler.jitter_queue = [r.clone() for r in rand] if training else None
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    ref = g["out"]
    out = model(rb)
    b = out["ray_samples"].flat_bins if training else model.sample_and_forward_field(model.collide(rb))["ray_samples"].flat_bins
    assert b.shape == ref["bins"].shape and (b[:, 1:] >= b[:, :-1]).all() and b.min() >= 0 and b.max() <= 1
    d_bins = (b.cpu() - ref["bins"]).abs()
    assert d_bins.median().item() <= 1e-4 and d_bins.max().item() <= 2e-2, (d_bins.median().item(), d_bins.max().item())
    assert_close("rgb (own samples)", out["rgb"], ref["rgb"], rtol=2e-2, atol=2e-2)
    # identical samples
    rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
    sdf, grad, rgb, x = model.field.forward_fused(rs)
    density = model.field.laplace_density(sdf)
    weights = density_to_weights(density, rs.flat_starts, rs.flat_ends)
    out_rgb = (weights[..., None] * rgb).sum(1)
    acc = weights.sum(1)
    mid = (rs.flat_starts + rs.flat_ends) / 2
    depth = ((weights * mid).sum(1) / (acc + 1e-10)).clip(mid.min(), mid.max())
    normal = (weights[..., None] * F.normalize(grad, p=2, dim=-1)).sum(1)
    assert_close("sdf", sdf, ref["sdf"], rtol=0, atol=1e-5)
    assert_close("gradient", grad, ref["gradient"], rtol=1e-4, atol=1e-6)
    assert_close("density", density, ref["density"], rtol=1e-4, atol=1e-5)
    assert_close("weights", weights, ref["weights"], rtol=1e-4, atol=1e-6)
    assert_close("rendered rgb", out_rgb if training else out_rgb.clamp(0, 1), ref["rgb"], rtol=1e-4, atol=1e-6)
    hit = ref["accumulation"] > 0.05
    assert_close("rendered depth", depth[hit.to(device)], ref["depth"][hit], rtol=1e-4, atol=1e-6)
    assert_close("rendered normal", normal, ref["normal"], rtol=1e-4, atol=1e-6)
    assert_close("accumulation", acc, ref["accumulation"], rtol=1e-4, atol=1e-6)
    if training:
        loss = F.l1_loss(g["in"]["image"].to(device), out_rgb) + ((grad.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
        assert_close("loss", loss, g["loss"]["rgb_loss"] + g["loss"]["eikonal_loss"], rtol=1e-4, atol=1e-7)
        model.zero_grad()
        loss.backward()
        got = product_grads(model)
        n_checked = 0
        for k, rg in g["grad"].items():
            assert k in got, f"no gradient for {k}"
            assert_close(f"grad {k}", got[k], rg, rtol=5e-3, atol=1e-8)
            n_checked += 1
        assert n_checked >= 40


# ------------------------------------------------------------------------------------------------ density weights
@pytest.mark.parametrize("S", [24, 96, 256])
def test_density_weights_fwd_bwd(device, S):
    from sdfstudio_amd.model_components.renderers import density_to_weights

    torch.manual_seed(2)
    n = 50
    starts = torch.sort(torch.rand(n, S + 1) * 4 + 0.5, dim=-1)[0]
    st, en = starts[:, :-1].contiguous(), starts[:, 1:].contiguous()
    dens = (torch.rand(n, S) * 30).requires_grad_(True)
    wref = O.weights_from_density(dens, en - st)
    coef = torch.randn(n, S)
    (wref * coef).sum().backward()
    dg = dens.detach().to(device).requires_grad_(True)
    w = density_to_weights(dg, st.to(device), en.to(device))
    (w * coef.to(device)).sum().backward()
    assert_close("weights", w, wref, rtol=1e-5, atol=1e-7)
    assert_close("density grad", dg.grad, dens.grad, rtol=1e-4, atol=1e-7)


# ------------------------------------------------------------------------------------------------ proposal field
def test_proposal_field_fwd_bwd(device):
    from sdfstudio_amd.fields.density_fields import HashMLPDensityField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    torch.manual_seed(3)
    pc = O.ProposalCfg(hidden_dim=16, num_levels=5, max_res=64, base_res=16, log2_hashmap_size=12)
    p = O.init_proposal_params([pc], seed=5)
    p["proposal_networks.0.table"] = (torch.rand_like(p["proposal_networks.0.table"]) * 2 - 1) * 0.5
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    n, s = 40, 24
    o, d, cam = O.synthetic_rays(n)
    starts = torch.sort(torch.rand(n, s + 1) * 5 + 0.3, dim=-1)[0]
    st, en = starts[:, :-1].contiguous(), starts[:, 1:].contiguous()
    mid = o[:, None, :] + d[:, None, :] * ((st + en) / 2)[..., None]
    ref = O.proposal_density(mid, p, "proposal_networks.0", pc)
    coef = torch.randn(n, s)
    (ref * coef).sum().backward()

    net = HashMLPDensityField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), spatial_distortion=SceneContraction(order=float("inf")),
                              hidden_dim=16, num_levels=5, max_res=64, base_res=16, log2_hashmap_size=12)
    with torch.no_grad():
        net.mlp_base.table.copy_(p["proposal_networks.0.table"])
        net.mlp_base.w1.copy_(p["proposal_networks.0.w1"])
        net.mlp_base.w2.copy_(p["proposal_networks.0.w2"])
    net = net.to(device)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    rs = rb.get_ray_samples(st.to(device), en.to(device))
    dens = net.density_fn(rs)[..., 0]
    (dens * coef.to(device)).sum().backward()
    assert_close("proposal density", dens, ref, rtol=1e-5, atol=1e-6)
    assert_close("proposal table grad", net.mlp_base.table.grad, p["proposal_networks.0.table"].grad, rtol=1e-4, atol=1e-7)
    assert_close("proposal w1 grad", net.mlp_base.w1.grad, p["proposal_networks.0.w1"].grad, rtol=1e-4, atol=1e-7)
    assert_close("proposal w2 grad", net.mlp_base.w2.grad, p["proposal_networks.0.w2"].grad, rtol=1e-4, atol=1e-7)
    # explicit-position form (Field.density_fn, base_field.py:48-65) agrees with the fused form
    dens2 = net.density_fn(mid.to(device))[..., 0]
    assert_close("density_fn(positions)", dens2, ref, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ renderer
@pytest.mark.parametrize("S", [16, 128])
@pytest.mark.parametrize("white", [False, True])
def test_neus_render_fwd_bwd(device, S, white):
    from sdfstudio_amd.model_components.renderers import neus_render

    torch.manual_seed(4)
    n = 33
    o, d, cam = O.synthetic_rays(n)
    starts = torch.sort(torch.rand(n, S + 1) * 4 + 0.5, dim=-1)[0]
    st, en = starts[:, :-1].contiguous(), starts[:, 1:].contiguous()
    t = (st + en) / 2
    sdf = (2.0 - t + 0.05 * torch.randn(n, S)).requires_grad_(True)  # crosses zero along the ray
    grad = (-d[:, None, :] + 0.3 * torch.randn(n, S, 3)).requires_grad_(True)
    rgb = torch.rand(n, S, 3, requires_grad=True)
    var = torch.tensor([0.35], requires_grad=True)
    bg = torch.ones(3) if white else None
    ca = 0.4
    alpha = O.neus_alpha(sdf, grad, d, en - st, O.neus_inv_s(var), ca)
    w, _ = O.weights_from_alphas(alpha)
    r_rgb, r_depth, r_normal, r_acc = O.render(w, rgb, torch.nn.functional.normalize(grad, dim=-1), st, en, bg)
    c1, c2, c3, c4, c5 = torch.randn(n, 3), torch.randn(n), torch.randn(n, 3), torch.randn(n), torch.randn(n, S)
    ((r_rgb * c1).sum() + (r_depth * c2).sum() + (r_normal * c3).sum() + (r_acc * c4).sum() + (w * c5).sum()).backward()

    g = lambda x: x.detach().to(device).requires_grad_(True)
    sdf_g, grad_g, rgb_g, var_g = g(sdf), g(grad), g(rgb), g(var)
    out_rgb, depth, normal, acc, weights, alpha_g = neus_render(
        sdf_g, grad_g, rgb_g, var_g, d.to(device), st.to(device), en.to(device), ca, None if bg is None else bg.to(device))
    dv = lambda x: x.to(device)
    ((out_rgb * dv(c1)).sum() + (depth * dv(c2)).sum() + (normal * dv(c3)).sum() + (acc * dv(c4)).sum()
     + (weights * dv(c5)).sum()).backward()
    assert_close("alpha", alpha_g, alpha, rtol=1e-5, atol=2e-6)
    assert_close("weights", weights, w, rtol=1e-5, atol=2e-6)
    assert_close("rgb", out_rgb, r_rgb, rtol=1e-4, atol=1e-6)
    assert_close("depth", depth, r_depth, rtol=1e-4, atol=1e-6)
    assert_close("normal", normal, r_normal, rtol=1e-4, atol=1e-6)
    assert_close("acc", acc, r_acc, rtol=1e-5, atol=1e-6)
    assert_close("sdf grad", sdf_g.grad, sdf.grad, rtol=2e-3, atol=1e-6)
    assert_close("gradient grad", grad_g.grad, grad.grad, rtol=2e-3, atol=1e-6)
    assert_close("rgb grad", rgb_g.grad, rgb.grad, rtol=1e-4, atol=1e-7)
    assert_close("variance grad", var_g.grad, var.grad, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("shape", [(128, 256, 96), (16, 32, 24), (48, 7, 5), (128, 64, 1)])
def test_interlevel_loss_kernel(device, shape):
    """sdfhip_interlevel_terms (one wave per ray, merge instead of sort) against the oracle's statement of losses.py:116-172:
    loss value and the gradient w.r.t. both proposal levels' weights; ragged sizes, empty-weight rays, coincident knots."""
    from sdfstudio_amd.model_components.losses import interlevel_loss_zip

    S, S0, S1 = shape
    n = 67
    torch.manual_seed(S + S0)
    def bins(s):
        b = torch.sort(torch.rand(n, s + 1), dim=-1)[0]
        b[:, 0], b[:, -1] = 0.0, 1.0
        return b
    cb = bins(S)
    w = torch.rand(n, S) * (torch.rand(n, S) < 0.5)
    w[5] = 0.0                                  # a ray that hit nothing
    w = w / w.sum(-1, keepdim=True).clamp_min(1e-3)
    bl = [bins(S0), bins(S1), cb]
    bl[0][7] = torch.linspace(0, 1, S0 + 1)     # proposal edges that coincide with nothing / everything
    wl = [torch.rand(n, S0).requires_grad_(True), torch.rand(n, S1).requires_grad_(True), w]
    ref = O.interlevel_loss_zip(wl, bl)
    ref.backward()
    wg = [x.detach().to(device).requires_grad_(True) for x in wl[:2]] + [w.to(device)]
    got = interlevel_loss_zip(wg, [b.to(device) for b in bl])
    got.backward()
    assert torch.isfinite(ref).item()
    # the blur divides differences of normalised weights by 2 r = 0.006 and the loss differences cumulative sums: fp32 round-off of
    # the reference's own sequential cumsum is ~1e-3 of single gradient entries, so the bar is the fp64 evaluation of the oracle
    # (the kernel must be as close to it as the fp32 oracle is, x3), plus the plain fp32 comparison relative to the maximum
    w64 = [x.detach().double().requires_grad_(True) for x in wl[:2]] + [w.double()]
    truth = O.interlevel_loss_zip(w64, [b.double() for b in bl])
    truth.backward()
    assert_close("interlevel loss", got, ref.detach(), rtol=1e-4, atol=1e-7)
    for i in (0, 1):
        assert_fp32_class(f"d / d prop weights {i}", wg[i].grad, wl[i].grad, w64[i].grad, factor=3.0,
                          atol=1e-5 * w64[i].grad.abs().max().item())
        assert_close(f"d / d prop weights {i} (vs fp32 oracle)", wg[i].grad, wl[i].grad, rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize("world", [1, 4])
def test_fused_adam_against_torch_adam(device, world):
    """sdfhip_adam_step over the flat buffers of two parameter groups (one misaligned slice, ragged sizes) against
    torch.optim.Adam with the reference's settings (eps 1e-15, per-group lr, NeuS warm-up / cosine schedule), gradients of widely
    different scales; with world > 1 the 1 / world mean rides in the kernel's gradient read."""
    from sdfstudio_amd.distributed import FlatGradients
    from sdfstudio_amd.engine.optimizers import Optimizers, neus_scheduler

    torch.manual_seed(1)
    shapes_a, shapes_b = [(37, 5), (3,), (1001,)], [(64, 7), (2,)]
    mk = lambda shp: [torch.nn.Parameter(torch.randn(*s_, device=device)) for s_ in shp]
    ga, gb = mk(shapes_a), mk(shapes_b)
    ref_a = [torch.nn.Parameter(p.detach().clone()) for p in ga]
    ref_b = [torch.nn.Parameter(p.detach().clone()) for p in gb]
    sched = neus_scheduler(3, 0.05, 10)
    opts = Optimizers({"fields": {"lr": 5e-4, "scheduler": sched}, "proposal_networks": {"lr": 1e-2, "scheduler": None}},
                      {"fields": ga, "field_background": [], "proposal_networks": gb})
    t_a = torch.optim.Adam(ref_a, lr=5e-4, eps=1e-15)
    t_b = torch.optim.Adam(ref_b, lr=1e-2, eps=1e-15)
    s_a = torch.optim.lr_scheduler.LambdaLR(t_a, sched)
    for step in range(6):
        opts.zero_grad_all()
        for p, r in zip(ga + gb, ref_a + ref_b):
            g = torch.randn_like(p) * 10.0 ** torch.randint(-7, 2, p.shape, device=device).float()
            opts.flat_grads._view(p).add_(g * world)  # what a SUM all-reduce over `world` identical ranks leaves in the flat buffer
            r.grad = g.clone()
        opts.optimizer_step_all(grad_scale=1.0 / world)
        opts.scheduler_step_all(step)
        t_a.step(), t_b.step(), s_a.step()
        for i, (p, r) in enumerate(zip(ga + gb, ref_a + ref_b)):
            assert_close(f"step {step} param {i}", p, r, rtol=2e-6, atol=1e-7)
    assert ga[0].data_ptr() == opts.adam.flat_params.flat.data_ptr()  # parameters really live in the flat buffer


def test_fused_adamw_groups_against_torch(device):
    """sdfhip_adamw_step: the optimizer dictionary of the reference's neuralangelo / bakedangelo presets (configs/method_configs.py:229-236,
    156-163) - `fields` on AdamW with DECOUPLED weight decay 0.01, `field_background` on AdamW with weight decay 0, both under the
    MultiStepWarmup schedule - handed over as the reference's config objects and stepped beside torch.optim.AdamW; plus a group on
    torch.optim.Adam's L2 weight decay.  Per-group decay: one launch per group over the flat buffers."""
    from sdfstudio_amd.engine.optimizers import (AdamOptimizerConfig, AdamWOptimizerConfig, MultiStepWarmupSchedulerConfig, Optimizers,
                                                 multi_step_warmup_scheduler)

    torch.manual_seed(2)
    mk = lambda shp: [torch.nn.Parameter(torch.randn(*s_, device=device)) for s_ in shp]
    ga, gb, gc = mk([(37, 5), (3,), (1001,)]), mk([(64, 7), (2,)]), mk([(129,), (5, 5)])
    refs = [[torch.nn.Parameter(p.detach().clone()) for p in g] for g in (ga, gb, gc)]
    sch = lambda: MultiStepWarmupSchedulerConfig(warm_up_end=3, milestones=[5, 8], gamma=0.1)
    opts = Optimizers({"fields": {"optimizer": AdamWOptimizerConfig(lr=1e-3, eps=1e-15, weight_decay=0.01), "scheduler": sch()},
                       "field_background": {"optimizer": AdamWOptimizerConfig(lr=1e-3, eps=1e-15), "scheduler": sch()},
                       "proposal_networks": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15, weight_decay=0.003), "scheduler": None}},
                      {"fields": ga, "field_background": gb, "proposal_networks": gc})
    assert opts.adam.groups["fields"]["decoupled"] and opts.adam.groups["fields"]["weight_decay"] == 0.01
    assert not opts.adam.groups["proposal_networks"]["decoupled"]
    t = [torch.optim.AdamW(refs[0], lr=1e-3, eps=1e-15, weight_decay=0.01), torch.optim.AdamW(refs[1], lr=1e-3, eps=1e-15, weight_decay=0),
         torch.optim.Adam(refs[2], lr=1e-2, eps=1e-15, weight_decay=0.003)]
    f = multi_step_warmup_scheduler(3, (5, 8), 0.1)
    ts = [torch.optim.lr_scheduler.LambdaLR(t[0], f), torch.optim.lr_scheduler.LambdaLR(t[1], f)]
    for step in range(10):
        opts.zero_grad_all()
        for p, r in zip(ga + gb + gc, refs[0] + refs[1] + refs[2]):
            g = torch.randn_like(p) * 10.0 ** torch.randint(-5, 1, p.shape, device=device).float()
            opts.flat_grads._view(p).add_(g)
            r.grad = g.clone()
        opts.optimizer_step_all()
        opts.scheduler_step_all(step)
        for o in t:
            o.step()
        for s_ in ts:
            s_.step()
        for i, (p, r) in enumerate(zip(ga + gb + gc, refs[0] + refs[1] + refs[2])):
            assert_close(f"step {step} param {i}", p, r, rtol=2e-6, atol=1e-7)


def test_fused_adam_skips_never_active_table_rows_exactly(device):
    """Progressive hash levels: FlatGradients.set_active_numel keeps the table rows of switched-off levels out of zero() and of the
    Adam step (FlatGradients.live_ranges).  torch.optim.Adam with explicit zero gradients there must end at the same parameters: the
    skipped rows stay bit-for-bit where they started, rows switched on later start from zero moments with the GLOBAL step count."""
    from sdfstudio_amd.engine.optimizers import Optimizers

    torch.manual_seed(2)
    head, table, tail = (torch.nn.Parameter(torch.randn(n, device=device)) for n in (33, 1000, 7))
    refs = [torch.nn.Parameter(p.detach().clone()) for p in (head, table, tail)]
    start = table.detach().clone()
    opts = Optimizers({"fields": {"lr": 1e-3, "scheduler": None}}, {"fields": [head, table, tail]})
    t = torch.optim.Adam(refs, lr=1e-3, eps=1e-15)
    for step, active in enumerate([301, 301, 301, 640, 640, 1000]):
        opts.flat_grads.set_active_numel(table, active)
        launches = len(opts.flat_grads.live_ranges())
        assert launches == (1 if active == 1000 else 2)
        opts.zero_grad_all()
        for p, r in zip((head, table, tail), refs):
            g = torch.randn_like(p)
            if p is table:
                g[active:] = 0.0
            opts.flat_grads._view(p).add_(g)
            r.grad = g.clone()
        opts.optimizer_step_all()
        t.step()
        for i, (p, r) in enumerate(zip((head, table, tail), refs)):
            assert_close(f"step {step} param {i}", p, r, rtol=2e-6, atol=1e-7)
        assert torch.equal(table.detach()[active:], start[active:])


# ------------------------------------------------------------------------------------------------ field (small golden net)
def _field_case(cfg, params, n, s, seed, use_emb=False):
    torch.manual_seed(seed)
    o, d, cam = O.synthetic_rays(n, seed=seed)
    starts = torch.sort(torch.rand(n, s) * 4.0 + 0.5, dim=-1)[0]
    return o, d, cam, starts


def _oracle_field(cfg_f, p, o, d, cam, starts, coefs, mask=None, training=True):
    po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
    n, s = starts.shape
    fo = O.field_outputs(o, d, starts, torch.ones(n, s, dtype=starts.dtype), cam, po, cfg_f, mask=mask, training=training)
    loss = (fo["sdf"] * coefs[0]).sum() + (fo["gradient"] * coefs[1]).sum() + (fo["rgb"] * coefs[2]).sum()
    loss.backward()
    return fo, po


def _product_field(model, o, d, cam, starts, coefs, device):
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    st = starts.to(device)
    rs = rb.get_ray_samples(st, st + 1.0)
    sdf, grad, rgb, x = model.field.forward_fused(rs)
    loss = (sdf * coefs[0].to(device)).sum() + (grad * coefs[1].to(device)).sum() + (rgb * coefs[2].to(device)).sum()
    model.zero_grad()
    loss.backward()
    return sdf, grad, rgb, x


FIELD_KEYS = ["glin0", "glin3", "glin4", "glin5", "glin8", "clin0", "clin2", "clin4"]


def _check_field_grads(model, po, rtol=1e-3, truth=None, min_checked=28, clin_rtol=5e-3):
    """Parameter gradients against the fp32 oracle (|d| <= rtol * max|ref|); with `truth` (the oracle evaluated in
    fp64) the bar is the fp32-round-off class of the reference path instead (helpers.assert_fp32_class)."""
    got = product_grads(model)
    checked = 0
    for k, ref in po.items():
        if ref.grad is None or k.startswith("proposal") or k.startswith("laplace") or k.startswith("deviation"):
            continue
        if k == "embedding_appearance.embedding.weight" and k not in got:
            continue
        assert k in got, f"no gradient produced for {k}"
        if truth is None:
            assert_close(f"grad {k}", got[k], ref.grad, rtol=rtol, atol=1e-9)
        else:
            # ReLU / clip masks make the gradient piecewise: one unit flipping at one point is a discrete jump, so the
            # bar is max(3 x the fp32 oracle's own distance from fp64, rtol x scale)
            # colour network: a single ReLU unit whose pre-activation the two fp32 evaluations put on opposite sides of zero moves one
            # row of a clin* gradient by a discrete amount (helpers.relu_flip_basis; up to ~1e-2 of the maximum at 8192 samples, see
            # test_full_shape_training_step_against_oracle): 5e-3 of the maximum for those tensors
            rt = max(rtol, clin_rtol) if k.startswith("clin") else rtol
            assert_fp32_class(f"grad {k}", got[k], ref.grad, truth[k].grad, factor=3.0, atol=rt * truth[k].grad.abs().max().item())
        checked += 1
    assert checked >= min_checked


@pytest.mark.parametrize("n,s", [(16, 8), (37, 13)])
def test_field_small_fwd_bwd(device, n, s):
    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).train()
    o, d, cam, starts = _field_case(cfg, g["param"], n, s, seed=11)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, g["param"], o, d, cam, starts, coefs)
    sdf, grad, rgb, x = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=1e-5)
    assert_close("gradient", grad, fo["gradient"], rtol=1e-4, atol=1e-5)
    assert_close("rgb", rgb, fo["rgb"], rtol=0, atol=2e-5)
    assert_close("points_norm", x.norm(dim=-1), fo["points_norm"], rtol=1e-6, atol=1e-6)
    _check_field_grads(model, po)


def test_field_small_expanded_cotangents(device):
    """Cotangents that arrive as stride-0 expands (plain .sum() losses with DIFFERENT scales per head): the binding must keep
    every contiguous copy it makes alive until the launch, or two of them alias one allocator block."""
    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).train()
    n, s = 16, 8
    o, d, cam, starts = _field_case(cfg, g["param"], n, s, seed=5)
    scales = (1.0, -0.25, 3.0)
    coefs = [torch.full((n, s), scales[0]), torch.full((n, s, 3), scales[1]), torch.full((n, s, 3), scales[2])]
    fo, po = _oracle_field(cfg.field, g["param"], o, d, cam, starts, coefs)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    st = starts.to(device)
    sdf, grad, rgb, _ = model.field.forward_fused(rb.get_ray_samples(st, st + 1.0))
    model.zero_grad()
    (scales[0] * sdf.sum() + scales[1] * grad.sum() + scales[2] * rgb.sum()).backward()  # all three cotangents are expands
    _check_field_grads(model, po)


def test_field_without_weight_norm(device):
    """SDFFieldConfig(weight_norm=False) (sdf_field.py:146, 312-313, 360-361): plain nn.Linear layers named glin{l}.weight / .bias.  The
    golden network with its weight-norm folded (W = g v / |v|) must give the same outputs, and the gradients arrive on .weight."""
    g = load_golden("train")
    cfg = small_oracle_cfg()
    plain = {}
    for k, v in g["param"].items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            plain[k[: -len("_v")]] = O.fold_weight_norm(v, g["param"][k[: -len("_v")] + "_g"])
        else:
            plain[k] = v
    model = product_model_from_params(plain, cfg, device, field_kwargs={"weight_norm": False}).train()
    names = dict(model.field.named_parameters())
    assert "glin0.weight" in names and "glin0.weight_v" not in names and "clin2.weight" in names
    n, s = 23, 9
    o, d, cam, starts = _field_case(cfg, plain, n, s, seed=21)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, plain, o, d, cam, starts, coefs)
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=1e-5)
    assert_close("gradient", grad, fo["gradient"], rtol=1e-4, atol=1e-5)
    assert_close("rgb", rgb, fo["rgb"], rtol=0, atol=2e-5)
    _check_field_grads(model, po, min_checked=19)


def test_field_small_level_mask_and_reference_style_outputs(device):
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).train()
    model.field.update_mask(5)  # progressive hash levels (sdf_field.py:376-378)
    mask = torch.ones(16)
    mask[10:] = 0
    n, s = 9, 6
    o, d, cam, starts = _field_case(cfg, g["param"], n, s, seed=12)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, g["param"], o, d, cam, starts, coefs, mask=mask)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    st = starts.to(device)
    rs = rb.get_ray_samples(st, st + 1.0)
    out = model.field(rs, return_alphas=True)
    assert_close("sdf (masked levels)", out[H.SDF][..., 0], fo["sdf"], rtol=0, atol=1e-5)
    assert_close("density", out[H.DENSITY][..., 0], fo["density"], rtol=1e-4, atol=1e-5)
    assert_close("normal", out[H.NORMAL], fo["normal"], rtol=1e-4, atol=1e-5)
    assert_close("alpha (get_alpha)", out[H.ALPHA][..., 0], fo["alpha"], rtol=1e-4, atol=1e-5)
    assert out["points_norm"].shape == (n, s, 1) and out["sampled_sdf"] is None
    loss = (out[H.SDF][..., 0] * coefs[0].to(device)).sum() + (out[H.GRADIENT] * coefs[1].to(device)).sum() + (
        out[H.RGB] * coefs[2].to(device)).sum()
    model.zero_grad()
    loss.backward()
    got = product_grads(model)
    assert_close("table grad (masked)", got["encoding.params"], po["encoding.params"].grad, rtol=1e-3, atol=1e-9)
    assert_close("glin0 grad (masked)", got["glin0.weight_v"], po["glin0.weight_v"].grad, rtol=1e-3, atol=1e-9)


def test_field_small_get_sdf_and_geonetwork(device):
    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).eval()
    n, s = 21, 5
    o, d, cam, starts = _field_case(cfg, g["param"], n, s, seed=13)
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    st = starts.to(device)
    rs = rb.get_ray_samples(st, st + 1.0)
    pos = (o[:, None, :] + d[:, None, :] * starts[..., None]).reshape(-1, 3)  # NOT contracted (sdf_field.py:412-418)
    with torch.no_grad():
        h = O.geo_network(pos, g["param"], cfg.field)
    assert_close("get_sdf", model.field.get_sdf(rs)[..., 0], h[:, 0].view(n, s), rtol=0, atol=1e-5)
    out = model.field.forward_geonetwork(pos.to(device))
    assert_close("forward_geonetwork sdf", out[:, 0], h[:, 0], rtol=0, atol=1e-5)
    assert_close("forward_geonetwork feat", out[:, 1:], h[:, 1:], rtol=1e-4, atol=1e-5)
    dens, feat = model.field.get_density(rs)  # sdf_field.py:469-475
    beta = g["param"]["laplace_density.beta"].abs() + g["param"]["laplace_density.beta_min"]
    assert_close("get_density", dens[..., 0], O.laplace_density(h[:, 0], beta).view(n, s), rtol=1e-4, atol=1e-4)
    assert_close("get_density feature", feat.reshape(n * s, -1), h[:, 1:], rtol=1e-4, atol=1e-5)


def test_geonetwork_positions_outside_the_grid_cube(device):
    """get_sdf / forward_geonetwork take UNcontracted positions (sdf_field.py:380-418) and the samplers feed them points outside
    the [-2, 2]^3 cube the hash grid covers; tiny-cuda-nn reduces the dense levels' index modulo the level size (grid.h
    grid_index), so such points read wrapped cells instead of faulting.  Same wrap in the kernels, value for value."""
    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).eval()
    gen = torch.Generator().manual_seed(5)
    pos = (torch.rand(777, 3, generator=gen) * 2 - 1) * 7.0
    pos[:5] = torch.tensor([[2.0, 2.0, 2.0], [-2.0, -2.0, -2.0], [2.0001, 0.0, 0.0], [-6.9, 6.9, -6.9], [0.0, 0.0, 0.0]])
    with torch.no_grad():
        h = O.geo_network(pos, g["param"], cfg.field)
    out = model.field.forward_geonetwork(pos.to(device))
    assert_close("forward_geonetwork sdf outside the cube", out[:, 0], h[:, 0], rtol=1e-5, atol=2e-5)
    assert_close("forward_geonetwork feat outside the cube", out[:, 1:], h[:, 1:], rtol=1e-4, atol=2e-5)
    model.train()
    gr = model.field.gradient(pos.to(device), skip_spatial_distortion=True)
    _, _, ref = O.sdf_and_gradient(pos, g["param"], cfg.field, create_graph=False)
    assert_close("analytic gradient outside the cube", gr, ref, rtol=1e-4, atol=5e-5)


def test_forward_geonetwork_is_differentiable(device):
    """SDFField.forward_geonetwork under autograd (sdf_field.py:380-410; first-order backward kernel, no tangent pass) against the
    oracle's geo_network: outputs and the gradients of every geometry-network weight and of the hash table; the colour network
    must receive exactly zero."""
    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).train()
    torch.manual_seed(21)
    n = 333  # ragged: 2 full workgroups + a tail
    pos = (torch.rand(n, 3) * 2 - 1) * 0.9
    c_sdf, c_feat = torch.randn(n), torch.randn(n, cfg.field.geo_feat_dim) * 0.1
    po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
    h = O.geo_network(pos, po, cfg.field)
    ((h[:, 0] * c_sdf).sum() + (h[:, 1:] * c_feat).sum()).backward()
    out = model.field.forward_geonetwork(pos.to(device))
    assert out.requires_grad
    assert_close("sdf", out[:, 0], h[:, 0], rtol=0, atol=1e-5)
    assert_close("feature", out[:, 1:], h[:, 1:], rtol=1e-4, atol=1e-5)
    model.zero_grad()
    ((out[:, 0] * c_sdf.to(device)).sum() + (out[:, 1:] * c_feat.to(device)).sum()).backward()
    got = product_grads(model)
    checked = 0
    for k, ref in po.items():
        if ref.grad is None:
            if k.startswith("clin") and k in got:
                assert got[k].abs().max().item() == 0.0, k
            continue
        if k.startswith("glin") or k == "encoding.params":
            assert_close(f"grad {k}", got[k], ref.grad, rtol=1e-3, atol=1e-9)
            checked += 1
    assert checked >= 9 * 3 + 1
    with torch.no_grad():
        lite = model.field.forward_geonetwork(pos.to(device))
    assert not lite.requires_grad
    assert_close("no-grad variant", lite, out, rtol=0, atol=0)


def test_field_appearance_embedding(device):
    g = load_golden("train")
    cfg = small_oracle_cfg()
    cfg.field.use_appearance_embedding = True
    model = product_model_from_params(g["param"], cfg, device).train()
    n, s = 12, 7
    o, d, cam, starts = _field_case(cfg, g["param"], n, s, seed=14)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, g["param"], o, d, cam, starts, coefs)
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("rgb (appearance)", rgb, fo["rgb"], rtol=0, atol=2e-5)
    got = product_grads(model)
    assert_close("embedding grad", got["embedding_appearance.embedding.weight"],
                 po["embedding_appearance.embedding.weight"].grad, rtol=1e-3, atol=1e-9)


# ------------------------------------------------------------------------------------------------ whole step vs golden
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_model_against_reference_golden(device, mode):
    g = load_golden(mode)
    cfg = small_oracle_cfg()
    training = mode == "train"
    model = product_model_from_params(g["param"], cfg, device).train(training)
    model.field.set_cos_anneal_ratio(float(g["in"]["cos_anneal"]))
    model.proposal_sampler.set_anneal(float(g["in"]["anneal"]))
    model.proposal_sampler.initial_sampler.jitter_override = g["in"]["rand0"].to(device)
    # the PDF sampler is called twice per forward with different draws: feed them in order
    draws = [g["in"]["rand1"].to(device), g["in"]["rand2"].to(device)]
    pdf = model.proposal_sampler.pdf_sampler
    orig = pdf.generate_ray_samples

    def patched(*a, **k):
        pdf.jitter_override = draws.pop(0) if draws else None
        return orig(*a, **k)

    pdf.generate_ray_samples = patched
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    out = model(rb)
    ref = g["out"]
    rs = out["ray_samples"] if training else None
    if training:
        assert_close("bins", rs.flat_bins, ref["bins"], rtol=0, atol=2e-5)
        assert_close("starts", rs.flat_starts, ref["starts"], rtol=1e-5, atol=2e-5)
        fo = out["field_outputs"]
        from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

        assert_close("sdf", fo[H.SDF][..., 0], ref["sdf"], rtol=0, atol=3e-5)
        assert_close("alpha", fo[H.ALPHA][..., 0], ref["alpha"], rtol=1e-3, atol=1e-4)
        assert_close("prop_weights0", out["weights_list"][0][..., 0], ref["prop_weights0"], rtol=1e-4, atol=1e-6)
        assert_close("prop_weights1", out["weights_list"][1][..., 0], ref["prop_weights1"], rtol=1e-4, atol=1e-5)
    assert_close("weights", out["weights"][..., 0], ref["weights"], rtol=1e-3, atol=1e-4)
    # end to end the sample positions come out of three inverse-CDF resamplings (see test_pdf_sampler for their fp32
    # conditioning); with inv_s = 20 and 16 samples per ray a 5e-5 shift of a sample moves its alpha by ~3e-4.  The 1e-4
    # bar on rendered rgb / depth is enforced on identical samples in test_field_and_render_on_reference_samples.
    assert_close("rgb", out["rgb"], ref["rgb"], rtol=5e-4, atol=1e-4)
    assert_close("accumulation", out["accumulation"][..., 0], ref["accumulation"], rtol=5e-4, atol=1e-4)
    # expected depth divides by the accumulated weight: compare where the ray actually hits something
    hit = ref["accumulation"] > 0.05
    assert_close("depth", out["depth"][..., 0][hit.to(device)], ref["depth"][hit], rtol=1e-4, atol=1e-4)
    assert_close("normal", out["normal"], ref["normal"], rtol=1e-3, atol=1e-4)
    if training:
        losses = model.get_loss_dict(out, {"image": g["in"]["image"]})
        for k, v in g["loss"].items():
            assert_close(f"loss {k}", losses[k], v, rtol=2e-4, atol=1e-7)
        model.zero_grad()
        sum(losses.values()).backward()
        got = product_grads(model)
        n_checked = 0
        for k, rg in g["grad"].items():
            assert k in got, f"no gradient for {k}"
            assert_close(f"grad {k}", got[k], rg, rtol=5e-3, atol=1e-8)
            n_checked += 1
        assert n_checked >= 40


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_field_and_render_on_reference_samples(device, mode):
    """north_star parity bar on IDENTICAL rays and samples: the reference's own sample positions (golden starts / ends)
    through the HIP field + renderer: 1e-5 on SDF, 1e-4 relative on rendered RGB / depth."""
    from sdfstudio_amd.model_components.renderers import neus_render

    g = load_golden(mode)
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).train(mode == "train")
    ca = float(g["in"]["cos_anneal"])
    model.field.set_cos_anneal_ratio(ca)
    ref = g["out"]
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
    with torch.no_grad():
        sdf, grad, rgb, x = model.field.forward_fused(rs)
        out_rgb, depth, normal, acc, weights, alpha = neus_render(
            sdf, grad, rgb, model.field.deviation_network.variance, rs.flat_directions, rs.flat_starts, rs.flat_ends, ca, None)
    if mode == "eval":
        out_rgb = out_rgb.clamp(0.0, 1.0)
    assert_close("sdf", sdf, ref["sdf"], rtol=0, atol=1e-5)
    assert_close("field rgb", rgb, ref["field_rgb"], rtol=1e-4, atol=1e-6)
    assert_close("gradient", grad, ref["gradient"], rtol=1e-4, atol=1e-6)
    assert_close("alpha", alpha, ref["alpha"], rtol=1e-4, atol=1e-6)
    assert_close("weights", weights, ref["weights"], rtol=1e-4, atol=1e-6)
    assert_close("rendered rgb", out_rgb, ref["rgb"], rtol=1e-4, atol=1e-6)
    hit = ref["accumulation"] > 0.05
    assert_close("rendered depth", depth[hit.to(device)], ref["depth"][hit], rtol=1e-4, atol=1e-6)
    assert_close("rendered normal", normal, ref["normal"], rtol=1e-4, atol=1e-6)
    assert_close("accumulation", acc, ref["accumulation"], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ full-size network
def test_field_full_size_fwd_bwd(device):
    """BASELINE config 2 network (16x2x2^19 grid, 8x256 + 4x256) on a few thousand points against the oracle."""
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3))
    gen = torch.Generator().manual_seed(21)
    p = O.init_field_params(cfg.field, seed=3)
    for k in list(p):
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=gen)
        elif k == "encoding.params":
            # 1/f spectrum: amplitude 0.3 at the coarsest level falling with the level's scale, so that every level
            # contributes a comparable d feature / d x (a flat 0.1 at scale 2e3 makes |d sdf/dx| ~ 13 and the fp32
            # reference path itself loses 3 digits there)
            lv = cfg.field.grid_levels()
            t = (torch.rand(p[k].shape, generator=gen) * 2 - 1).view(-1, 2)
            for l in range(lv.n_levels):
                t[int(lv.offset[l]):int(lv.offset[l + 1])] *= 0.3 * float(lv.scale[0]) / float(lv.scale[l])
            p[k] = t.reshape(-1)
    p.update(O.init_proposal_params(cfg.proposals))
    cfg_small_props = cfg
    model = product_model_from_params(p, cfg_small_props, device).train()
    n, s = 40, 50  # 2000 points: not a multiple of 128 -> exercises the padded tail
    o, d, cam, starts = _field_case(cfg, p, n, s, seed=15)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, p, o, d, cam, starts, coefs)
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("sdf (8x256)", sdf, fo["sdf"], rtol=0, atol=1e-5)
    # d sdf/dx and rgb go through the finest hash levels (scale ~2e3, table amplitude 0.1 here): the fp32 oracle is itself
    # only 2e-4 (relative) from its fp64 evaluation, so the bar is "same round-off class as the fp32 reference path"
    f64, p64 = _oracle_field(cfg.field, to_double(p), o.double(), d.double(), cam, starts.double(), [c.double() for c in coefs])
    assert_fp32_class("gradient (8x256)", grad, fo["gradient"], f64["gradient"], factor=3.0, atol=2e-5)
    assert_fp32_class("rgb (8x256)", rgb, fo["rgb"], f64["rgb"], factor=3.0, atol=2e-5)
    _check_field_grads(model, po, rtol=1e-3, truth=p64)


@pytest.mark.parametrize("nl", [8, 2, 5])
def test_field_hidden_512_layer_by_layer(device, nl):
    """Hidden width 512 (neus-facto-bigmlp, method_configs.py:503-523: num_layers 8, hidden_dim 512, colour 4 x 256; also a shallow
    network without the skip connection and a 5-layer one whose skip layer is the last hidden layer): two 16-block accumulator sets do
    not fit a wave, the geometry network runs layer by layer (csrc/wide_kernels.h) on the per-layer tensors of the training data flow.
    Same bars as the fused shapes: sdf 1e-5, everything else fp64-anchored; plus the inference entries (get_sdf, forward_geonetwork,
    the no-grad forward) that run the same launches without a backward."""
    fcfg = O.FieldCfg(num_layers=nl, hidden_dim=512, num_layers_color=4, bias=0.5, inside_outside=False, beta_init=0.3)
    cfg = O.ModelCfg(field=fcfg)
    p = _full_shape_params(cfg, seed=31 + nl)
    model = product_model_from_params(p, cfg, device).train()
    n, s = 33, 40
    o, d, cam, starts = _field_case(cfg, p, n, s, seed=23)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, p, o, d, cam, starts, coefs)
    sdf, grad, rgb, x = _product_field(model, o, d, cam, starts, coefs, device)
    tag = f"({nl}x512 + 4x256)"
    assert_close(f"sdf {tag}", sdf, fo["sdf"], rtol=0, atol=1e-5)
    f64, p64 = _oracle_field(cfg.field, to_double(p), o.double(), d.double(), cam, starts.double(), [c.double() for c in coefs])
    assert_fp32_class(f"gradient {tag}", grad, fo["gradient"], f64["gradient"], factor=3.0, atol=2e-5)
    assert_fp32_class(f"rgb {tag}", rgb, fo["rgb"], f64["rgb"], factor=3.0, atol=2e-5)
    _check_field_grads(model, po, rtol=1e-3, truth=p64, min_checked=2 * (nl + 1) + 2 * 5, clin_rtol=3e-2)
    # inference entries: nothing differentiated, same launches
    model.eval()
    rb = _bundle(o, d, cam, 0.5, 4.5, device)
    st = starts.to(device)
    rs = rb.get_ray_samples(st, st + 1.0)
    with torch.no_grad():
        sdf_e, grad_e, rgb_e, _ = model.field.forward_fused(rs)
    assert_close("eval-mode sdf == training sdf", sdf_e, sdf.detach(), rtol=0, atol=0)
    assert_close("eval-mode gradient", grad_e, grad.detach(), rtol=0, atol=0)
    pos = (o[:, None, :] + d[:, None, :] * starts[..., None]).reshape(-1, 3)
    with torch.no_grad():
        h = O.geo_network(pos, p, cfg.field)
    assert_close("get_sdf", model.field.get_sdf(rs)[..., 0], h[:, 0].view(n, s), rtol=0, atol=1e-5)
    out = model.field.forward_geonetwork(pos.to(device))
    assert_close("forward_geonetwork feat", out[:, 1:], h[:, 1:], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 2, 256), (1, 1, 256), (5, 2, 256), (6, 3, 256), (7, 4, 256), (9, 5, 256), (5, 1, 64), (2, 3, 64), (9, 2, 64)],
                         ids=lambda t: f"{t[0]}x{t[2]}+{t[1]}x{t[2]}")
def test_field_depth_sweep_fwd_bwd(device, shape):
    """Depth is a run-time property of the fused kernels (they loop over the layers): (geometry layers, colour layers, width)
    swept against the oracle, fp64-anchored - forward outputs and every parameter gradient.  (2, 2, 256) is the `neus-facto`
    PRESET's field (method_configs.py:472-480: no skip connection), depths >= 5 have the skip connection at layer 4
    (sdf_field.py:280,300-305), 9 is the deepest the argument tables hold; 256-wide shapes run on the 16 x 2 x 2^19 grid of
    BASELINE config 2, 64-wide ones on the small golden grid."""
    nl, nlc, width = shape
    if width == 256:
        fcfg = O.FieldCfg(num_layers=nl, num_layers_color=nlc, bias=0.5, inside_outside=False, beta_init=0.3)
    else:
        base = small_oracle_cfg().field
        fcfg = O.FieldCfg(**{**base.__dict__, "num_layers": nl, "num_layers_color": nlc})
    cfg = O.ModelCfg(field=fcfg) if width == 256 else small_oracle_cfg()
    cfg.field = fcfg
    p = _full_shape_params(cfg, seed=5 + nl)
    model = product_model_from_params(p, cfg, device).train()
    n, s = 33, 40
    o, d, cam, starts = _field_case(cfg, p, n, s, seed=19)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    fo, po = _oracle_field(cfg.field, p, o, d, cam, starts, coefs)
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    tag = f"({nl}x{width} + {nlc}x{width})"
    assert_close(f"sdf {tag}", sdf, fo["sdf"], rtol=0, atol=1e-5)
    f64, p64 = _oracle_field(cfg.field, to_double(p), o.double(), d.double(), cam, starts.double(), [c.double() for c in coefs])
    assert_fp32_class(f"gradient {tag}", grad, fo["gradient"], f64["gradient"], factor=3.0, atol=2e-5)
    assert_fp32_class(f"rgb {tag}", rgb, fo["rgb"], f64["rgb"], factor=3.0, atol=2e-5)
    # 1320 points: ONE colour-network ReLU unit that the two fp32 evaluations put on opposite sides of zero at ONE point changes an
    # entry of a clin* gradient by one term of a ~1320-term sum of random-sign terms, i.e. by a few percent of the tensor's maximum
    # (seen: 1.9 % on clin4.weight_v of the 9 + 5 shape, with the fp32 oracle 2e-6 from fp64).  The geometry network (Softplus, no
    # knife edges) keeps the 1e-3 bar; a wrong colour kernel would be off by O(1) in every tensor and in rgb above.
    _check_field_grads(model, po, rtol=1e-3, truth=p64, min_checked=2 * (nl + 1) + 2 * (nlc + 1), clin_rtol=3e-2)


def test_full_size_properties(device):
    """BASELINE config 2 shape end to end (4096 rays would take the oracle minutes; use size-independent properties)."""
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3))
    p = O.init_field_params(cfg.field, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    model = product_model_from_params(p, cfg, device).train()
    n = 1024
    o, d, cam = O.synthetic_rays(n)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    w = out["weights"][..., 0]
    assert w.shape == (n, 128) and (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all()
    rs = out["ray_samples"]
    assert (rs.flat_starts[:, 1:] >= rs.flat_starts[:, :-1]).all()
    assert rs.flat_starts.min() >= cfg.near - 1e-4 and rs.flat_ends.max() <= cfg.far + 1e-4
    g = out["eik_grad"]
    inside = out["points_norm"][..., 0] < 0.9
    assert ((g.norm(dim=-1) - 1).abs()[inside]).mean() < 0.3  # eikonal ~ 1 at geometric init (statistical)
    assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    sum(losses.values()).backward()
    for k, prm in model.named_parameters():
        # laplace_density.beta feeds only the (unused here) DENSITY head: NeuS renders from alpha (neus.py:94-104)
        if prm.requires_grad and "embedding" not in k and "laplace_density" not in k:
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k


def test_config1_volsdf_pure_mlp_full_size(device):
    """BASELINE config 1 shape: VolSDF, pure-MLP field (8x256 + 4x256, zero grid features), 512 rays x (64 + 32) samples,
    ErrorBoundedSampler with 128 evaluation samples (up to 640 merged samples per ray).  Size-independent properties."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig

    torch.manual_seed(0)
    fcfg = SDFFieldConfig(bias=0.5, inside_outside=False, use_grid_feature=False, beta_init=0.1)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    model = VolSDFModel(VolSDFModelConfig(sdf_field=fcfg, background_model="none"), box, num_train_data=49).to(device).train()
    n = 512
    o, d, cam = O.synthetic_rays(n)
    out = model(_bundle(o, d, cam, 0.5, 4.5, device))
    rs = out["ray_samples"]
    assert rs.flat_starts.shape == (n, 96)
    assert (rs.flat_bins[:, 1:] >= rs.flat_bins[:, :-1]).all() and rs.flat_starts.min() >= 0.5 - 1e-4 and rs.flat_ends.max() <= 4.5 + 1e-4
    w = out["weights"][..., 0]
    assert (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all() and torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()
    # geometric init = sphere of radius 0.5: rays through the volume must accumulate, and the expected depth sits near the sphere
    hit = out["accumulation"][:, 0] > 0.9
    assert hit.float().mean() > 0.2
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    sum(losses.values()).backward()
    for k, prm in model.named_parameters():
        if prm.requires_grad and "embedding" not in k and "encoding" not in k and "deviation" not in k:
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k


def test_config4_inside_out_mono_priors(device):
    """BASELINE config 4 shape: NeuS-facto in an inside-out (room) scene, `inside_outside=True` (sign-flipped geometric init,
    sdf_field.py:294-299), cameras INSIDE the box, monocular depth + normal priors (README.md:74: mono-depth-loss-mult 0.1,
    mono-normal-loss-mult 0.05).  Same kernels as config 2; size-independent properties of the step."""
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.8, inside_outside=True, beta_init=0.3))
    p = O.init_field_params(cfg.field, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    model = product_model_from_params(p, cfg, device).train()
    model.config.mono_depth_loss_mult, model.config.mono_normal_loss_mult = 0.1, 0.05
    n = 1024  # N = 0 mod 32: the depth loss reshapes to (1, 32, -1) (base_surface_model.py:427-437)
    gen = torch.Generator().manual_seed(4)
    o = (torch.rand(n, 3, generator=gen) - 0.5) * 0.6  # camera centres inside the room
    d = F.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    cam = torch.randint(0, 49, (n,), generator=gen)
    out = model(_bundle(o, d, cam, 0.05, 4.0, device))
    w = out["weights"][..., 0]
    assert (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all()
    # inside-out init: sdf = bias - |x| is positive at the cameras, every ray leaves the sphere of radius `bias` -> it hits
    assert (out["accumulation"][:, 0] > 0.9).float().mean() > 0.9
    depth = out["depth"][:, 0]
    assert torch.isfinite(depth).all() and (depth > 0.05).all() and (depth < 1.5).all()
    batch = {"image": torch.rand(n, 3, generator=gen), "depth": torch.rand(n, generator=gen),
             "normal": F.normalize(torch.randn(n, 3, generator=gen), dim=-1)}
    losses = model.get_loss_dict(out, batch)
    assert {"rgb_loss", "eikonal_loss", "depth_loss", "normal_loss", "interlevel_loss"} <= set(losses), sorted(losses)
    for k, v in losses.items():
        assert torch.isfinite(v), k
    sum(losses.values()).backward()
    for k, prm in model.named_parameters():
        if prm.requires_grad and "embedding" not in k and "laplace_density" not in k:
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k


def test_numerical_gradient_field_against_reference_golden(device):
    """use_numerical_gradients (neus-facto-angelo's field mode, sdf_field.py:431-453,638-644): the geometry network at the samples
    and at six taps each in one differentiable call, finite-difference normals, the colour network on them, `sampled_sdf`, and
    the gradients of rgb-L1 + eikonal + curvature (neus_facto.py:312-325) against the reference's own run."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

    g = load_golden_file("numgrad_small_train.npz")
    cfg = small_oracle_cfg()
    i, ref = g["in"], g["out"]
    delta, curv_mult = float(i["delta"]), float(i["curv_mult"])
    n, s = i["starts"].shape
    model = product_model_from_params(g["param"], cfg, device).train()
    fld = model.field
    fld.config.use_numerical_gradients = True
    fld.set_numerical_gradients_delta(delta)
    rb = _bundle(i["origins"], i["dirs"], i["cam"], cfg.near, cfg.far, device)
    rs = rb.get_ray_samples(i["starts"].to(device), i["ends"].to(device))
    out = fld(rs)
    assert_close("sdf", out[H.SDF][..., 0], ref["sdf"], rtol=0, atol=1e-5)
    assert_close("sampled_sdf", out["sampled_sdf"], ref["sampled_sdf"], rtol=0, atol=1e-5)
    # the finite difference divides the sdf's fp32 round-off (~1e-6) by 2 delta = 1 / 64
    assert_close("gradient", out[H.GRADIENT], ref["gradient"], rtol=1e-4, atol=1e-4)
    assert_close("normal", out[H.NORMAL], ref["normal"], rtol=1e-4, atol=1e-4)
    assert_close("rgb", out[H.RGB], ref["field_rgb"], rtol=1e-4, atol=2e-5)
    gd, taps = fld.gradient(fld_positions := (i["origins"][:, None, :] + i["dirs"][:, None, :] * i["starts"][..., None]).to(device),
                            return_sdf=True)  # sdf_field.py:424: contracts, then the same six taps
    assert_close("gradient()", gd, ref["gradient"], rtol=1e-4, atol=1e-4)
    assert taps.shape == (6, n, s)
    image = i["image"].to(device)
    curvature = (out["sampled_sdf"].reshape(n, s, 3, 2).sum(dim=-1) - 2 * out[H.SDF]) / (delta * delta)
    losses = {"rgb_loss": F.l1_loss(image, out[H.RGB]),
              "eikonal_loss": ((out[H.GRADIENT].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult,
              "curvature_loss": curvature.abs().mean() * curv_mult}
    for k, v in g["loss"].items():
        assert_close(f"loss {k}", losses[k], v, rtol=5e-4, atol=1e-7)
    model.zero_grad()
    sum(losses.values()).backward()
    got = product_grads(model)

    def oracle_backward():
        po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
        o = O.field_outputs(i["origins"], i["dirs"], i["starts"], i["ends"] - i["starts"], i["cam"], po, cfg.field, None, 1.0, True,
                            numerical_delta=delta)
        curv = ((o["sampled_sdf"].reshape(n, s, 3, 2).sum(-1) - 2 * o["sdf"][..., None]) / (delta * delta)).abs().mean() * curv_mult
        (F.l1_loss(o["rgb"], i["image"]) + ((o["gradient"].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult + curv).backward()
        return {k: po[k].grad for k in g["grad"]}

    for k in g["grad"]:
        assert k in got, f"no gradient for {k}"
    assert len(g["grad"]) >= 40
    _, basis = relu_flip_basis(oracle_backward, margin=3e-5)  # the finite-difference normal carries ~1e-4 of noise into the colour net
    assert_grads_close_mod_relu_flips(got, g["grad"], basis, rtol=5e-3)


def test_numerical_gradient_model_step_properties(device):
    """neus-facto-angelo flavour of the step at BASELINE config 2's network shape: numerical gradients with the delta of the
    coarsest active level, progressive level mask, curvature loss on `sampled_sdf` (method_configs.py:381-450,
    neus_facto.py:209-225,312-325).  Size-independent properties of one training step."""
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.5, inside_outside=False, beta_init=0.3))
    p = O.init_field_params(cfg.field, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    model = product_model_from_params(p, cfg, device).train()
    fld = model.field
    fld.config.use_numerical_gradients = True
    model.config.curvature_loss_multi = 5e-4
    level = 8
    fld.update_mask(level)                                                               # sdf_field.py:376-378
    fld.set_numerical_gradients_delta(1.0 / (fld.base_res * fld.growth_factor ** (level - 1)))  # neus_facto.py:219-222
    n = 512
    o, d, cam = O.synthetic_rays(n)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    fo = out["field_outputs"]
    assert fo["sampled_sdf"].shape == (n, 128, 6) and torch.isfinite(fo["sampled_sdf"]).all()
    w = out["weights"][..., 0]
    assert (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all() and torch.isfinite(out["rgb"]).all()
    g = out["eik_grad"]
    inside = out["points_norm"][..., 0] < 0.9
    assert ((g.norm(dim=-1) - 1).abs()[inside]).mean() < 0.3  # finite-difference normal of the sphere init is still ~ unit
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    assert "curvature_loss" in losses and torch.isfinite(losses["curvature_loss"]) and losses["curvature_loss"] > 0
    sum(losses.values()).backward()
    table_grad = None
    for k, prm in model.named_parameters():
        if prm.requires_grad and "embedding" not in k and "laplace_density" not in k:
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
        if k.endswith("encoding.params") and k.startswith("field"):
            table_grad = prm.grad
    # masked levels (>= level) receive exactly zero gradient (what lets DDP skip them)
    first_masked = fld.encoding.levels[level].offset
    assert table_grad.view(-1, 2)[first_masked:].abs().max().item() == 0.0
    # (at geometric init the first layer ignores the grid features, sdf_field.py:296-299, so the active levels' gradient is
    # zero too; the non-trivial table gradients are checked against the reference in the golden test above)


def _angelo_field_cfg(log2_t):
    """neus-facto-angelo's field (method_configs.py:403-422): 16 levels x 8 features, linear interpolation, base 64 -> 4096,
    one 256-wide hidden layer, no positional encoding (zeroed), appearance embedding, numerical gradients."""
    return O.FieldCfg(num_layers=1, hidden_dim=256, geo_feat_dim=256, num_layers_color=4, hidden_dim_color=256, bias=0.5,
                      inside_outside=False, beta_init=0.3, use_appearance_embedding=True, use_position_encoding=False,
                      num_levels=16, max_res=4096, base_res=64, log2_hashmap_size=log2_t, hash_features_per_level=8,
                      hash_smoothstep=False)


def test_config5_shape_field_fwd_bwd(device):
    """BASELINE config 5's field SHAPE (8 features per level, linear interpolation, 1-hidden-layer geometry network, in0 = 167) on
    a 2^16 table: the analytic path and the preset's own numerical-gradient path against the oracle, anchored on its fp64
    evaluation.  (The analytic normal through the 4095-scale linear level is what exposed a last-bit difference in the level
    scale between libm's exp2f and numpy's: both sides now evaluate it in double, oracle/hashgrid.py make_levels.)"""
    fc = _angelo_field_cfg(16)
    cfg = O.ModelCfg(field=fc)
    gen = torch.Generator().manual_seed(31)
    p = O.init_field_params(fc, seed=5)
    lv = fc.grid_levels()
    t = (torch.rand(p["encoding.params"].shape, generator=gen) * 2 - 1).view(-1, 8)
    for l in range(lv.n_levels):  # 1/f spectrum (see test_field_full_size_fwd_bwd)
        t[int(lv.offset[l]):int(lv.offset[l + 1])] *= 0.2 * float(lv.scale[0]) / float(lv.scale[l])
    p["encoding.params"] = t.reshape(-1)
    for k in list(p):
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=gen)
    p.update(O.init_proposal_params(cfg.proposals))
    model = product_model_from_params(p, cfg, device).train()
    n, s = 24, 20  # 480 points: padded tail
    o, d, cam, starts = _field_case(cfg, p, n, s, seed=16)
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    # ---- analytic d sdf / dx (one fused call, second-order backward)
    fo, po = _oracle_field(fc, p, o, d, cam, starts, coefs)
    f64, p64 = _oracle_field(fc, to_double(p), o.double(), d.double(), cam, starts.double(), [c.double() for c in coefs])
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=1e-5)
    assert_fp32_class("gradient", grad, fo["gradient"], f64["gradient"], factor=3.0, atol=2e-5)
    assert_fp32_class("rgb", rgb, fo["rgb"], f64["rgb"], factor=3.0, atol=2e-5)
    _check_field_grads(model, po, rtol=1e-3, truth=p64, min_checked=22)
    # ---- numerical gradients (the preset's mode), delta of level 8 (neus_facto.py:219-222)
    delta = 1.0 / (fc.base_res * fc.growth_factor() ** 7)
    model.field.config.use_numerical_gradients = True
    model.field.set_numerical_gradients_delta(delta)

    def oracle_num(pp, oo, dd, ss, cc):
        pq = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in pp.items()}
        f = O.field_outputs(oo, dd, ss, torch.ones(n, s, dtype=ss.dtype), cam, pq, fc, None, 1.0, True, numerical_delta=delta)
        ((f["sdf"] * cc[0]).sum() + (f["gradient"] * cc[1]).sum() + (f["rgb"] * cc[2]).sum()).backward()
        return f, pq

    fo, po = oracle_num(p, o, d, starts, coefs)
    f64, p64 = oracle_num(to_double(p), o.double(), d.double(), starts.double(), [c.double() for c in coefs])
    sdf, grad, rgb, _ = _product_field(model, o, d, cam, starts, coefs, device)
    assert_close("sdf (numerical mode)", sdf, fo["sdf"], rtol=0, atol=1e-5)
    assert_fp32_class("gradient (numerical)", grad, fo["gradient"], f64["gradient"], factor=3.0, atol=1e-5 / delta)
    assert_fp32_class("rgb (numerical)", rgb, fo["rgb"], f64["rgb"], factor=3.0, atol=5e-5)
    _check_field_grads(model, po, rtol=2e-3, truth=p64, min_checked=22)  # 2 geometry + 5 colour layers x 3, table, embedding


def test_config5_full_shape_step_properties(device):
    """BASELINE config 5 at its real sizes where the path is built (2048 rays x 48 samples, 16 x 8 x 2^22 table = 2.1 GB, numerical
    gradients, progressive levels from level_init = 8, curvature loss; background model "grid" is out of scope, far = 4.5)."""
    fc = _angelo_field_cfg(22)
    cfg = O.ModelCfg(field=fc)
    cfg.num_neus_samples = 48
    p = O.init_field_params(fc, seed=0)
    p.update(O.init_proposal_params(cfg.proposals))
    model = product_model_from_params(p, cfg, device, field_kwargs={"use_numerical_gradients": True}).train()
    model.config.curvature_loss_multi = 5e-4
    fld = model.field
    fld.update_mask(8)
    fld.set_numerical_gradients_delta(1.0 / (fld.base_res * fld.growth_factor ** 7))
    n = 2048
    o, d, cam = O.synthetic_rays(n)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    w = out["weights"][..., 0]
    assert w.shape == (n, 48) and (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all()
    assert out["field_outputs"]["sampled_sdf"].shape == (n, 48, 6)
    assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    assert {"rgb_loss", "eikonal_loss", "interlevel_loss", "curvature_loss"} <= set(losses)
    sum(losses.values()).backward()
    for k, prm in model.named_parameters():
        if prm.requires_grad and "laplace_density" not in k:
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
    tg = dict(model.named_parameters())["field.encoding.params"].grad.view(-1, 8)
    assert tg[fld.encoding.levels[8].offset:].abs().max().item() == 0.0  # masked levels: exactly zero (DDP can skip them)


def test_l2_scene_contraction(device):
    """scene_contraction_norm = "l2" (SceneContraction(order=None), base_surface_model.py:150-151, spatial_distortions.py:66-73):
    SDF field forward + backward and the proposal density on points well outside the unit ball, against the oracle."""
    from sdfstudio_amd.fields.density_fields import HashMLPDensityField
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneContraction

    g = load_golden("train")
    cfg = small_oracle_cfg()
    fc = cfg.field
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
                          num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
                          inside_outside=fc.inside_outside, use_grid_feature=True, beta_init=fc.beta_init, num_levels=fc.num_levels,
                          max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
                          hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep)
    fld = fcfg.setup(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=49, spatial_distortion=SceneContraction(order=None))
    from helpers import load_params

    class _Wrap(torch.nn.Module):
        def __init__(self, f):
            super().__init__()
            self.field = f
            self.proposal_networks = torch.nn.ModuleList()

    wrap = _Wrap(fld)
    load_params(wrap, {k: v for k, v in g["param"].items() if not k.startswith("proposal_networks.")})
    wrap = wrap.to(device).train()
    n, s = 29, 11
    o, d, cam = O.synthetic_rays(n, seed=5)
    starts = torch.sort(torch.rand(n, s) * 6.0 + 0.5, dim=-1).values  # up to |x| ~ 4: most samples are contracted
    coefs = [torch.randn(n, s), torch.randn(n, s, 3) * 0.3, torch.randn(n, s, 3)]
    po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
    fo = O.field_outputs(o, d, starts, torch.ones(n, s), cam, po, fc, contraction="l2")
    ((fo["sdf"] * coefs[0]).sum() + (fo["gradient"] * coefs[1]).sum() + (fo["rgb"] * coefs[2]).sum()).backward()
    sdf, grad, rgb, x = _product_field(wrap, o, d, cam, starts, coefs, device)
    assert_close("points_norm (L2-contracted)", x.norm(dim=-1), fo["points_norm"], rtol=1e-6, atol=1e-6)
    assert (x.norm(dim=-1) < 2.0).all() and (x.norm(dim=-1) > 1.0).float().mean() > 0.3
    assert_close("sdf", sdf, fo["sdf"], rtol=0, atol=1e-5)
    assert_close("gradient", grad, fo["gradient"], rtol=1e-4, atol=1e-5)
    assert_close("rgb", rgb, fo["rgb"], rtol=0, atol=2e-5)
    got = product_grads(wrap)
    for k in ("glin0.weight_v", "glin4.weight_v", "glin8.weight_v", "clin0.weight_v", "encoding.params"):
        assert_close(f"grad {k}", got[k], po[k].grad, rtol=1e-3, atol=1e-9)
    # proposal density with the same contraction
    pc = cfg.proposals[0]
    net = HashMLPDensityField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), spatial_distortion=SceneContraction(order=None),
                              hidden_dim=pc.hidden_dim, num_levels=pc.num_levels, max_res=pc.max_res, base_res=pc.base_res,
                              log2_hashmap_size=pc.log2_hashmap_size, features_per_level=pc.features_per_level).to(device)
    with torch.no_grad():
        net.mlp_base.table.copy_(g["param"]["proposal_networks.0.table"])
        net.mlp_base.w1.copy_(g["param"]["proposal_networks.0.w1"])
        net.mlp_base.w2.copy_(g["param"]["proposal_networks.0.w2"])
    pos = (o[:, None, :] + d[:, None, :] * starts[..., None])
    dens = net.density_fn(pos.to(device))
    ref = O.proposal_density(pos, g["param"], "proposal_networks.0", pc, contraction="l2")
    assert_close("proposal density (L2)", dens[..., 0], ref, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ background models
def _load_reference_state_dict(model, params):
    """Reference model.state_dict() -> our mirror: identical keys except the proposal networks' hash tables (the tinycudann shim
    keeps them as mlp_base.encoding.params) and two reference-only entries (device_indicator_param, proposal aabb)."""
    sd = model.state_dict()
    for k, v in params.items():
        if k == "device_indicator_param" or (k.startswith("proposal_networks.") and k.endswith(".aabb")):
            continue
        key = k.replace("mlp_base.encoding.params", "mlp_base.table") if k.startswith(("proposal_networks.", "field_background.")) else k
        assert key in sd, f"{key} missing from the mirror's state_dict"
        assert tuple(sd[key].shape) == tuple(v.shape), (key, tuple(sd[key].shape), tuple(v.shape))
        sd[key] = v.clone()
    missing = [k for k in sd if k not in {kk.replace("mlp_base.encoding.params", "mlp_base.table") for kk in params}]
    assert not missing, f"mirror parameters the reference checkpoint does not provide: {missing}"
    model.load_state_dict(sd)


@pytest.mark.parametrize("name", ["neus", "volsdf", "neus_facto", "neus_facto_grid", "neus_grid"])
def test_background_mlp_models_against_reference_golden(device, name):
    """background_model="mlp" (the reference's default) and "grid" (BASELINE config 5: TCNNNerfactoField, fields/nerfacto_field.py):
    NeuS / VolSDF add transmittance x colour of the samples beyond the far plane (base_surface_model.py:314-329), NeuS-facto merges
    the background field into alpha / colour outside the unit sphere (:266-290).  Golden: the reference's own model classes run end
    to end (tests/golden/make_golden_bg.py), eval mode; compared: rendered outputs, the rgb loss, and its gradient w.r.t. SDF
    field, background field and proposal networks."""
    import functools

    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.models import background as BG
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus import NeuSModel, NeuSModelConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig

    grid = name.endswith("_grid")
    bgm = "grid" if grid else "mlp"
    name = name[:-5] if grid else name
    g = load_golden_file(f"{name}_bg_{bgm}_eval.npz")
    if grid:  # the golden shrinks the background table through the field's own constructor arguments (make_golden_bg.py)
        full = BG.TCNNNerfactoField
        BG.TCNNNerfactoField = functools.partial(TCNNNerfactoField, num_levels=6, max_res=64, log2_hashmap_size=10)
    fcfg = SDFFieldConfig(num_layers=8, hidden_dim=64, geo_feat_dim=64, num_layers_color=4, hidden_dim_color=64, bias=0.5,
                          inside_outside=False, use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4,
                          log2_hashmap_size=11, hash_features_per_level=2, hash_smoothstep=True)
    props = [{"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 32, "base_res": 4},
             {"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 64, "base_res": 4}]
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    if name == "neus":
        model = NeuSModel(NeuSModelConfig(sdf_field=fcfg, background_model=bgm, num_samples=16, num_samples_importance=16,
                                          num_up_sample_steps=2, num_samples_outside=8), box, 49)
    elif name == "volsdf":
        model = VolSDFModel(VolSDFModelConfig(sdf_field=fcfg, background_model="mlp", num_samples=16, num_samples_eval=32,
                                              num_samples_extra=8, num_samples_outside=8), box, 49)
    else:
        model = NeuSFactoModel(NeuSFactoModelConfig(sdf_field=fcfg, background_model=bgm, num_proposal_samples_per_ray=(32, 24),
                                                    num_neus_samples_per_ray=16, proposal_net_args_list=props, num_samples_outside=8), box, 49)
    if grid:
        BG.TCNNNerfactoField = full
    _load_reference_state_dict(model, g["param"])
    model = model.to(device).eval()
    i = g["in"]
    n = i["origins"].shape[0]
    rb = RayBundle(origins=i["origins"].to(device), directions=i["directions"].to(device), directions_norm=torch.ones(n, 1, device=device),
                   camera_indices=i["camera_indices"][:, None].to(device))
    out = model(rb)
    loss = torch.nn.functional.l1_loss(i["image"].to(device), out["rgb"])
    model.zero_grad()
    loss.backward()
    o = g["out"]
    # NeuS re-samples four times from sdf-derived pdfs: one fp32 ulp of a cdf moves a bin edge by ~5e-6 and the next level
    # amplifies it (same 5e-3 bar as test_neus_model_against_reference_golden); the other two models follow the reference's
    # samples to ~1e-6
    tol = 5e-3 if name == "neus" else 5e-4
    # END-TO-END comparison (both sides run their own samplers): sample positions differ by fp32 round-off, so single elements move
    # by more than the tensor-level bars suggest and the element-wise gate of helpers.report does not apply (inf); element-wise
    # parity of the background fields on IDENTICAL samples: test_nerf_background_field_fwd_bwd, test_nerfacto_background_field_fwd_bwd
    E = float("inf")
    assert_close("weights", out["weights"], o["weights"], rtol=tol, atol=2e-6, elem_rtol=E)
    assert_close("rgb", out["rgb"], o["rgb"], rtol=tol, atol=1e-6, elem_rtol=E)
    assert_close("depth", out["depth"], o["depth"], rtol=tol, atol=1e-5, elem_rtol=E)
    assert_close("normal", out["normal"], o["normal"], rtol=tol, atol=1e-5, elem_rtol=E)
    assert_close("accumulation", out["accumulation"], o["accumulation"], rtol=tol, atol=1e-6, elem_rtol=E)
    assert_close("rgb_loss", loss, g["loss"]["rgb_loss"], rtol=1e-4, atol=1e-7)
    got = {k.replace("mlp_base.table", "mlp_base.encoding.params") if k.startswith(("proposal_networks.", "field_background.")) else k: p.grad
           for k, p in model.named_parameters() if p.grad is not None}
    checked = 0
    for k, ref in g["grad"].items():
        assert k in got, f"no gradient for {k}"
        # What this test pins is the background path: the background field's own gradients (they see the foreground only through
        # the transmittance / the inside-sphere mask) at 2e-3 of the tensor's maximum.  The SDF field's gradients come from 48
        # rays here: a single colour-network ReLU whose pre-activation the two fp32 evaluations place on opposite sides of zero
        # (|z| ~ 1e-6, helpers.relu_flip_basis) moves single entries by up to ~1e-2 of the maximum; their tight comparison is the
        # job of the train-mode goldens (test_*_model_against_reference_golden), here they get the loose bar.
        # (the hash table's finest level is the extreme case: an entry sees a handful of samples, its gradient here is ~1e-4)
        # background: 2e-3 on the heads / head MLP; the 8 x 256 base MLP's gradients (scale ~5e-5 here) get 1e-2
        rt = (1e-2 if "mlp_base" in k else 2e-3) if k.startswith("field_background") else (1e-1 if k == "field.encoding.params" else 2e-2)
        assert_close(f"grad {k}", got[k], ref, rtol=rt, atol=1e-9, elem_rtol=E)
        checked += 1
    assert checked >= (44 if grid else 50)
    assert any(k.startswith("field_background.mlp_base") for k in g["grad"])


def test_dense_grid_sdf_for_mesh_extraction(device):
    """scripts/extract_mesh.py:94-133 / utils/marching_cubes.py: sdf_on_grid (ray layout, sdf row only) and sdf_on_points against
    the oracle's forward_geonetwork on a lattice; the coarse-to-fine pyramid must reproduce the dense evaluation wherever the
    surface can be (|sdf| below the final threshold) while evaluating a fraction of the points."""
    from sdfstudio_amd.utils.marching_cubes import evaluate_crop_pyramid, sdf_on_grid, sdf_on_points

    g = load_golden("train")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).eval()
    lo, hi, res = (-0.9, -0.8, -0.7), (0.9, 0.85, 0.8), (13, 9, 21)
    vol = sdf_on_grid(model.field, lo, hi, res, chunk_points=1000)  # several x slabs
    ax = [torch.linspace(lo[a], hi[a], res[a]) for a in range(3)]
    pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    ref = O.geo_network(pts, g["param"], cfg.field)[:, 0].reshape(res)
    assert_close("sdf_on_grid", vol, ref, rtol=0, atol=1e-5)
    assert_close("sdf_on_points", sdf_on_points(model.field, pts.to(device), chunk=700), ref.reshape(-1), rtol=0, atol=1e-5)
    n = 32
    ax = [torch.linspace(-1.0, 1.0, n, device=device)] * 3
    cube = torch.stack(torch.meshgrid(*ax, indexing="ij"), 0)
    z, counts, fine = evaluate_crop_pyramid(lambda p: sdf_on_points(model.field, p), cube, 2.0)
    dense = sdf_on_points(model.field, cube.reshape(3, -1).T.contiguous())
    assert fine.any() and counts[-1] == int(fine.sum()) < n ** 3 and counts[0] == (n // 8) ** 3
    assert_close("pyramid == dense where it refined to full resolution", z[fine], dense[fine], rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------------ full BASELINE shapes vs the oracle
def _full_shape_params(cfg, seed=3):
    """BASELINE config 2 networks with every path alive: noise on the weight directions, a 1/f table spectrum (every hash level
    contributes a comparable d feature / d x: see test_field_full_size_fwd_bwd), trained-looking proposal tables."""
    gen = torch.Generator().manual_seed(seed + 40)
    p = O.init_field_params(cfg.field, seed=seed)
    for k in list(p):
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=gen)
        elif k == "encoding.params":
            lv = cfg.field.grid_levels()
            t = (torch.rand(p[k].shape, generator=gen) * 2 - 1).view(-1, cfg.field.hash_features_per_level)
            for l in range(lv.n_levels):
                t[int(lv.offset[l]):int(lv.offset[l + 1])] *= 0.3 * float(lv.scale[0]) / float(lv.scale[l])
            p[k] = t.reshape(-1)
    p.update(O.init_proposal_params(cfg.proposals))
    for k in list(p):
        if k.startswith("proposal_networks") and k.endswith(".table"):
            p[k] = (torch.rand(p[k].shape, generator=gen) * 2 - 1) * 0.5
    return p


def _inject_facto_draws(model, rand, device):
    model.proposal_sampler.initial_sampler.jitter_override = rand[0].to(device)
    draws = [rand[1].to(device), rand[2].to(device)]
    pdf = model.proposal_sampler.pdf_sampler
    orig = pdf.generate_ray_samples

    def patched(*a, **k):
        pdf.jitter_override = draws.pop(0) if draws else None
        return orig(*a, **k)

    pdf.generate_ray_samples = patched


@pytest.mark.parametrize("config", [2, 4])
def test_full_shape_training_step_against_oracle(device, config):
    """BASELINE config 2 (and config 4: inside-out scene + monocular depth / normal priors) at its FULL network and sampling
    shape - 16 x 2 x 2^19 smoothstep grid, 8 x 256 geometry + 4 x 256 colour MLP, 256 / 96 proposal samples -> 128 field samples
    per ray - on 64 rays, end to end through the product model (samplers -> field -> compositing -> losses -> every parameter
    gradient) against the oracle on the same rays and draws.  Gradients are anchored on the oracle's fp64 evaluation."""
    from sdfstudio_amd.model_components.losses import monosdf_depth_loss
    from oracle.sdf_path import monosdf_normal_loss

    inside = config == 4
    cfg = O.ModelCfg(field=O.FieldCfg(bias=0.8 if inside else 0.5, inside_outside=inside, beta_init=0.3), num_neus_samples=128,
                     near=0.05 if inside else 0.5, far=4.0 if inside else 4.5)
    p = _full_shape_params(cfg)
    model = product_model_from_params(p, cfg, device).train()
    n = 64  # N = 0 mod 32: the depth prior reshapes the batch to (1, 32, -1)
    gen = torch.Generator().manual_seed(17)
    if inside:
        o = (torch.rand(n, 3, generator=gen) - 0.5) * 0.6
        d = F.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        cam = torch.randint(0, 49, (n,), generator=gen)
        model.config.mono_depth_loss_mult, model.config.mono_normal_loss_mult = 0.1, 0.05
    else:
        o, d, cam = O.synthetic_rays(n, seed=6)
    image = torch.rand(n, 3, generator=gen)
    rand = [torch.rand(n, 1, generator=gen) for _ in range(3)]
    batch = {"image": image}
    if inside:
        batch.update({"depth": torch.rand(n, generator=gen), "normal": F.normalize(torch.randn(n, 3, generator=gen), dim=-1)})
    cos_anneal, anneal = 0.4, 0.8

    def oracle(dtype):
        cast = (lambda t: t.to(dtype) if t.is_floating_point() else t)
        po = {k: cast(v).clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
        out = O.neus_facto_forward(cast(o), cast(d), cam, po, cfg, anneal=anneal, cos_anneal_ratio=cos_anneal, rand=[cast(r) for r in rand],
                                   training=True)
        losses = O.neus_facto_loss(out, cast(image), cfg)
        if inside:  # the prior losses are the host functions themselves (pinned on the reference by the CPU tests)
            losses["normal_loss"] = monosdf_normal_loss(out["normal"], cast(batch["normal"])) * 0.05
            losses["depth_loss"] = monosdf_depth_loss(out["depth"][:, None], cast(batch["depth"])[..., None]) * 0.1
        sum(losses.values()).backward()
        return out, losses, po

    ref, ref_losses, po = oracle(torch.float32)
    _, _, p64 = oracle(torch.float64)
    model.field.set_cos_anneal_ratio(cos_anneal)
    model.proposal_sampler.set_anneal(anneal)
    _inject_facto_draws(model, rand, device)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

    assert out["ray_samples"].flat_starts.shape == (n, 128)
    assert_close("bins", out["ray_samples"].flat_bins, ref["bins"], rtol=0, atol=5e-5)
    assert_close("prop_weights0", out["weights_list"][0][..., 0], ref["weights_list"][0], rtol=1e-4, atol=1e-6)
    assert_close("prop_weights1", out["weights_list"][1][..., 0], ref["weights_list"][1], rtol=1e-3, atol=1e-5)
    assert_close("sdf", out["field_outputs"][H.SDF][..., 0], ref["field"]["sdf"], rtol=0, atol=1e-4)
    assert_close("weights", out["weights"][..., 0], ref["weights"], rtol=2e-3, atol=2e-4)
    assert_close("rgb", out["rgb"], ref["rgb"], rtol=1e-3, atol=2e-4)
    assert_close("accumulation", out["accumulation"][..., 0], ref["accumulation"], rtol=1e-3, atol=2e-4)
    hit = ref["accumulation"] > 0.05
    assert_close("depth", out["depth"][..., 0][hit.to(device)], ref["depth"][hit], rtol=5e-4, atol=2e-4)
    assert_close("normal", out["normal"], ref["normal"], rtol=2e-3, atol=2e-4)
    losses = model.get_loss_dict(out, batch)
    assert set(losses) == set(ref_losses), (sorted(losses), sorted(ref_losses))
    for k, v in ref_losses.items():
        assert_close(f"loss {k}", losses[k], v.detach(), rtol=1e-3, atol=1e-6)
    model.zero_grad()
    sum(losses.values()).backward()
    got = product_grads(model)
    checked = 0
    for k, rg in po.items():
        if rg.grad is None or k not in got or "embedding" in k:
            continue
        # as close to the fp64 evaluation as the fp32 oracle is (x3), or 5e-3 of the tensor's maximum: the samples of the two
        # fp32 paths differ by ~1e-5 after three resamplings, which the finest hash levels (scale 2e3) turn into percent-level
        # changes of single table-entry gradients
        # (colour network: ReLU units whose pre-activation the two fp32 paths put on opposite sides of zero move single rows by
        # up to ~1e-2 of the maximum at 8192 samples, helpers.relu_flip_basis)
        frac = 1e-2 if k.startswith("clin") else 5e-3
        assert_fp32_class(f"grad {k}", got[k], rg.grad, p64[k].grad, factor=3.0, atol=frac * p64[k].grad.abs().max().item())
        checked += 1
    assert checked >= 40


def test_config1_full_shape_volsdf_against_oracle(device):
    """BASELINE config 1 at its full shape (VolSDF, pure-MLP 8 x 256 + 4 x 256 field with zeroed grid features, ErrorBoundedSampler
    64 + 32 final samples out of up to 640 merged evaluations) on 96 rays in eval mode (deterministic sampler): samples, rendered
    outputs and every field gradient of the rgb loss against the oracle."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.volsdf import VolSDFModel, VolSDFModelConfig
    from helpers import load_params

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
    image = torch.rand(n, 3, generator=gen)
    po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
    ref = O.volsdf_forward(o, d, cam, po, cfg, rand=None, training=False)
    ref_loss = F.l1_loss(ref["rgb"].clamp(0, 1), image)
    ref_loss.backward()
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    b = model.sample_and_forward_field(model.collide(_bundle(o, d, cam, cfg.near, cfg.far, device)))["ray_samples"].flat_bins
    assert b.shape == ref["bins"].shape == (n, 97)
    d_bins = (b.cpu() - ref["bins"]).abs()
    assert d_bins.median().item() <= 1e-5 and d_bins.max().item() <= 2e-2, (d_bins.median().item(), d_bins.max().item())
    assert_close("rgb", out["rgb"], ref["rgb"].clamp(0, 1), rtol=2e-3, atol=5e-4)
    assert_close("accumulation", out["accumulation"][..., 0], ref["accumulation"], rtol=2e-3, atol=5e-4)
    hit = ref["accumulation"] > 0.05
    assert_close("depth", out["depth"][..., 0][hit.to(device)], ref["depth"][hit], rtol=2e-3, atol=1e-3)
    loss = F.l1_loss(out["rgb"], image.to(device))
    assert_close("rgb_loss", loss, ref_loss.detach(), rtol=1e-3, atol=1e-6)
    model.zero_grad()
    loss.backward()
    got = product_grads(model)
    checked = 0
    for k, rg in po.items():
        if rg.grad is None or k not in got or "encoding" in k or "embedding" in k or "deviation" in k:
            continue
        assert_close(f"grad {k}", got[k], rg.grad, rtol=2e-2, atol=1e-9)
        checked += 1
    assert checked >= 40


# ------------------------------------------------------------------------------------------------ UniSurf (surface root finder)
@pytest.mark.parametrize("training", [True, False])
def test_unisurf_sampler_against_oracle(device, training):
    """UniSurfSampler (ray_samplers.py:947-1138: marching samples -> sdf -> importance + outside samples -> first outside-to-inside
    sign change, interpolated depth, shrunk interval -> interval samples -> euclidean merge) on a geometric-init field (a sphere
    of radius 0.5: about half of the perturbed rays cross it) against the oracle (pinned on the reference's sampler by
    test_unisurf_sampler_oracle_against_reference)."""
    from sdfstudio_amd.model_components.ray_samplers import UniSurfSampler

    cfg = small_oracle_cfg()
    params = O.init_field_params(cfg.field, num_images=49, seed=3)
    g = {"param": params}
    model = product_model_from_params(params, cfg, device).train(training)
    n, M, K, Oo, I = 57, 64, 12, 9, 20
    torch.manual_seed(1)
    o, d, cam = O.synthetic_rays(n, seed=13)
    d = F.normalize(d + 0.15 * torch.randn(n, 3), dim=-1)
    smp = UniSurfSampler(num_samples_interval=I, num_samples_outside=Oo, num_samples_importance=K, num_marching_steps=M).train(training)
    smp.step_cb(4000)
    draws = [torch.rand(n, M + 1), torch.rand(n, K + 1), torch.rand(n, Oo + 1), torch.rand(n, I + 1)]
    smp.jitter_overrides = [t.to(device) for t in draws]
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    rs, sp = smp(rb, occupancy_fn=model.field.get_occupancy, sdf_fn=model.field.get_sdf, return_surface_points=True)
    nears, fars = torch.full((n,), cfg.near), torch.full((n,), cfg.far)
    ref = O.unisurf_sampler(nears, fars, lambda st: O.geo_network((o[:, None, :] + d[:, None, :] * st[..., None]).reshape(-1, 3), g["param"],
                                                                 cfg.field)[:, 0].view(st.shape),
                            lambda s: torch.sigmoid(-10.0 * s), smp.delta, draws if training else None, I, Oo, K, M)
    assert 0 < int(ref["mask"].sum()) < n, "the case must contain rays with and without a surface crossing"
    assert rs.flat_bins.shape == ref["bins"].shape
    # importance samples come out of an inverse CDF (5e-6 conditioning, see test_pdf_sampler); everything else is exact arithmetic
    assert_close("merged euclidean bins", rs.flat_bins, ref["bins"], rtol=0, atol=5e-5)
    ref_sp = o[ref["mask"]] + d[ref["mask"]] * ref["z"][ref["mask"]][:, None]
    assert_close("surface points", sp, ref_sp, rtol=0, atol=2e-5)
    assert torch.equal(rs.flat_starts[:, 1:], rs.flat_ends[:, :-1])


def test_analytic_gradient_on_points_and_unisurf_step(device):
    """SDFField.gradient(x) in analytic mode (sdf_field.py:455-467) against the oracle's autograd normal - with and without the
    scene contraction - and one UniSurf training step (unisurf.py:92-134: occupancy compositing, normal smoothness loss at the
    surface points) for finite losses and parameter gradients."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from sdfstudio_amd.models.unisurf import UniSurfModel, UniSurfModelConfig
    from helpers import load_params

    g = load_golden("train")
    cfg = small_oracle_cfg()
    fc = cfg.field
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim, num_layers_color=fc.num_layers_color,
                          hidden_dim_color=fc.hidden_dim_color, bias=fc.bias, inside_outside=fc.inside_outside, use_grid_feature=True,
                          beta_init=fc.beta_init, num_levels=fc.num_levels, max_res=fc.max_res, base_res=fc.base_res,
                          log2_hashmap_size=fc.log2_hashmap_size, hash_features_per_level=fc.hash_features_per_level,
                          hash_smoothstep=fc.hash_smoothstep)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = UniSurfModel(UniSurfModelConfig(sdf_field=fcfg, num_samples_interval=16, num_samples_importance=8, num_marching_steps=48,
                                            num_samples_outside=8, background_model="none"), box, 49)
    load_params(model, {k: v for k, v in g["param"].items() if not k.startswith("proposal_networks")})
    model = model.to(device).train()
    x = (torch.rand(301, 3) * 2 - 1) * 1.6  # some points outside the unit cube: the contraction matters
    for skip in (False, True):
        got = model.field.gradient(x.to(device), skip_spatial_distortion=skip)
        xin = x if skip else O.contract_inf(x)
        _, _, ref = O.sdf_and_gradient(xin, g["param"], fc, create_graph=False)
        assert_close(f"gradient(x), skip_spatial_distortion={skip}", got, ref, rtol=1e-4, atol=2e-5)
    n = 64
    o, d, cam = O.synthetic_rays(n, seed=2)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    w = out["weights"][..., 0]
    assert (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all() and out["ray_samples"].flat_starts.shape == (n, 32)
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    assert {"rgb_loss", "eikonal_loss", "normal_smoothness_loss"} <= set(losses)
    assert all(torch.isfinite(v) for v in losses.values()) and float(losses["normal_smoothness_loss"]) > 0
    model.zero_grad()
    sum(losses.values()).backward()
    for k in ("field.glin0.weight_v", "field.glin8.weight_v", "field.clin0.weight_v", "field.encoding.params"):
        gr = dict(model.named_parameters())[k].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().max() > 0, k
    model.after_train_iteration(100)
    assert model.sampler.delta < 0.25


# ------------------------------------------------------------------------------------------------ packed-sample path (NeuS-acc)
@pytest.mark.parametrize("res,step", [(16, 0.05), (32, 0.013)])
def test_occupancy_grid_marching(device, res, step):
    """sdfhip_march_count / _write (nerfacc.cuda.ray_marching as called at ray_samplers.py:1474-1484) against the fp32 oracle:
    sample counts, ray indices and interval ends bit for bit, on a random occupancy grid with rays that miss the box, graze it
    and cross it (parity unpinned: nerfacc is absent, the oracle restates its published algorithm)."""
    from sdfstudio_amd.model_components.ray_samplers import march_occupancy_grid

    gen = torch.Generator().manual_seed(res)
    binary = torch.rand(res, res, res, generator=gen) > 0.6
    n = 96
    o, d, _ = O.synthetic_rays(n, seed=7)
    d = F.normalize(d + 0.25 * torch.randn(n, 3, generator=gen), dim=-1)
    t_min, t_max = torch.full((n,), 0.5), torch.full((n,), 4.5)
    t_max[:7] = 0.4  # empty intervals
    roi = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    info, ri, ts, te = O.ray_marching(o, d, t_min, t_max, roi, binary, step)
    g_info, g_counts, g_ri, g_ts, g_te = march_occupancy_grid(o.to(device), d.to(device), t_min.to(device), t_max.to(device), roi,
                                                              binary.to(device), step)
    assert int(info[:, 1].sum()) > 500 and int((info[:, 1] == 0).sum()) >= 7
    assert torch.equal(g_info.cpu(), info) and torch.equal(g_counts.cpu().long(), info[:, 1])
    assert torch.equal(g_ri.cpu(), ri)
    assert torch.equal(g_ts.cpu(), ts) and torch.equal(g_te.cpu(), te)


def test_packed_weights_and_accumulate_fwd_bwd(device):
    """nerfacc.render_weight_from_alpha / accumulate_along_rays (models/neus_acc.py:103-121) on packed samples: segments of 0, 1,
    63..200 samples (several 64-sample rounds), alphas that are exactly 0 and exactly 1, forward and backward against autograd
    through the oracle's definitions."""
    from sdfstudio_amd.model_components.renderers import accumulate_along_rays, render_weight_from_alpha

    gen = torch.Generator().manual_seed(4)
    counts = torch.tensor([0, 1, 63, 64, 65, 0, 200, 17, 128, 0, 5], dtype=torch.int64)
    offs = torch.cumsum(counts, 0) - counts
    info = torch.stack([offs, counts], -1)
    P = int(counts.sum())
    ri = torch.repeat_interleave(torch.arange(len(counts)), counts)
    alpha = torch.rand(P, generator=gen) * 0.2
    alpha[torch.rand(P, generator=gen) < 0.05] = 0.0
    alpha[int(offs[6]) + 150] = 1.0  # ends ray 6 early
    values = torch.randn(P, 3, generator=gen)
    co = [torch.randn(len(counts), 3, generator=gen), torch.randn(len(counts), 1, generator=gen), torch.randn(P, generator=gen)]

    a_ref = alpha.clone().double().requires_grad_(True)
    v_ref = values.clone().double().requires_grad_(True)
    w_ref = O.packed_weights_from_alpha(a_ref, info)
    out_ref = O.accumulate_along_rays(w_ref, ri, v_ref, len(counts))
    acc_ref = O.accumulate_along_rays(w_ref, ri, None, len(counts))
    ((out_ref * co[0].double()).sum() + (acc_ref * co[1].double()).sum() + (w_ref * co[2].double()).sum()).backward()

    a = alpha.clone().to(device).requires_grad_(True)
    v = values.clone().to(device).requires_grad_(True)
    g_info, g_counts, g_ri = info.to(device), counts.to(torch.int32).to(device), ri.to(device)
    w = render_weight_from_alpha(a, g_info, g_counts)
    out = accumulate_along_rays(w, g_ri, v, g_info, g_counts)
    acc = accumulate_along_rays(w, g_ri, None, g_info, g_counts)
    ((out * co[0].to(device)).sum() + (acc * co[1].to(device)).sum() + (w * co[2].to(device)).sum()).backward()
    assert_close("packed weights", w, w_ref.float(), rtol=1e-5, atol=1e-7)
    assert_close("accumulated values", out, out_ref.float(), rtol=1e-5, atol=1e-6)
    assert_close("accumulation", acc, acc_ref.float(), rtol=1e-5, atol=1e-6)
    assert_close("d / d alpha", a.grad, a_ref.grad.float(), rtol=1e-4, atol=1e-6)
    assert_close("d / d values", v.grad, v_ref.grad.float(), rtol=1e-5, atol=1e-7)
    assert float(w[int(offs[6]) + 151:int(offs[6]) + 200].abs().max()) == 0.0  # nothing passes an alpha of 1


def test_neus_acc_model_packed_path(device):
    """NeuSAccModel (models/neus_acc.py:92-143) after its first occupancy-grid update (ray_samplers.py:1383-1432), on a geometric-init
    field (a sphere of radius 0.5): the pruned grid against the oracle's update rule, the packed samples against the oracle's march,
    rendered rgb / depth / normal / accumulation against the oracle's field + compositing on those samples, and one backward."""
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_acc import NeuSAccModel, NeuSAccModelConfig
    from sdfstudio_amd.models.neus_facto import SceneBox
    from helpers import load_params

    cfg = small_oracle_cfg()
    fc = cfg.field
    params = O.init_field_params(fc, num_images=49, seed=3)
    params["deviation_network.variance"] = torch.tensor([0.5])  # a trained-looking sharpness: inv_s = e^5, march step 5.9e-3
    fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim, num_layers_color=fc.num_layers_color,
                          hidden_dim_color=fc.hidden_dim_color, bias=fc.bias, inside_outside=fc.inside_outside, use_grid_feature=True,
                          beta_init=fc.beta_init, num_levels=fc.num_levels, max_res=fc.max_res, base_res=fc.base_res,
                          log2_hashmap_size=fc.log2_hashmap_size, hash_features_per_level=fc.hash_features_per_level,
                          hash_smoothstep=fc.hash_smoothstep)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
    model = NeuSAccModel(NeuSAccModelConfig(sdf_field=fcfg, num_samples=16, num_samples_importance=16, num_up_sample_steps=2, background_model="none"), box, 49)
    load_params(model, params)
    model = model.to(device).train()
    smp = model.sampler
    assert smp.resolution == 128 and int(smp._update_counter) == 0
    model.before_train_iteration(2000)  # step size from the variance (:1379-1382)
    inv_s = O.neus_inv_s(params["deviation_network.variance"])
    assert abs(smp.step_size - 14.0 / float(inv_s) / 16) < 1e-9
    model.after_train_iteration(2000)   # first grid update
    assert int(smp._update_counter) == 1
    ref_binary = O.neus_acc_binary_update(torch.ones(128, 128, 128, dtype=torch.bool), smp.cube_coordinate.cpu(),
                                          lambda x: O.geo_network(x, params, fc)[:, 0], inv_s, smp.voxel_size, smp.step_size)
    occ = smp._binary.cpu()
    assert 0.001 < float(ref_binary.float().mean()) < 0.5, float(ref_binary.float().mean())
    assert float((occ != ref_binary).float().mean()) < 1e-4  # voxels within round-off of the threshold may differ

    n = 64
    o, d, cam = O.synthetic_rays(n, seed=13)
    d = F.normalize(d + 0.1 * torch.randn(n, 3), dim=-1)
    out = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    info, ri, ts, te = O.ray_marching(o, d, torch.full((n,), cfg.near), torch.full((n,), cfg.far), torch.tensor([-1.0, -1, -1, 1, 1, 1]), occ,
                                      smp.step_size)
    assert ri.shape[0] > 100 and torch.equal(out["ray_indices"].cpu(), ri)
    assert torch.equal(out["ray_samples"].flat_starts.cpu(), ts) and torch.equal(out["ray_samples"].flat_ends.cpu(), te)
    with torch.no_grad():
        fo = O.field_outputs(o[ri], d[ri], ts, te - ts, cam[ri], params, fc, cos_anneal_ratio=model.field._cos_anneal_ratio, training=True)
    w = O.packed_weights_from_alpha(fo["alpha"][:, 0], info)
    assert_close("packed weights", out["packed_weights"][:, 0], w, rtol=1e-4, atol=1e-6)
    assert_close("rgb", out["rgb"], O.accumulate_along_rays(w, ri, fo["rgb"][:, 0], n), rtol=1e-4, atol=1e-5)
    assert_close("normal", out["normal"], O.accumulate_along_rays(w, ri, fo["normal"][:, 0], n), rtol=1e-4, atol=1e-5)
    assert_close("accumulation", out["accumulation"], O.accumulate_along_rays(w, ri, None, n), rtol=1e-4, atol=1e-5)
    assert_close("depth", out["depth"], O.accumulate_along_rays(w, ri, (ts + te) / 2, n), rtol=1e-4, atol=1e-5)
    losses = model.get_loss_dict(out, {"image": torch.rand(n, 3)})
    model.zero_grad()
    sum(losses.values()).backward()
    for k in ("field.glin0.weight_v", "field.clin0.weight_v", "field.deviation_network.variance"):
        gr = dict(model.named_parameters())[k].grad
        assert gr is not None and torch.isfinite(gr).all() and gr.abs().max() > 0, k
    # (the hash table's gradient is exactly zero here: the geometric initialisation zeroes layer 0's columns over the grid features)
    assert torch.isfinite(dict(model.named_parameters())["field.encoding.params"].grad).all()
    # rays that hit nothing: all-zero outputs, still a valid training step
    far_bundle = _bundle(o + 10.0, d, cam, cfg.near, cfg.far, device)
    out0 = model(far_bundle)
    assert float(out0["rgb"].abs().max()) == 0.0 and out0["eik_grad"].shape == (n, 3)
    # importance sampling (:1489-1500): 16 samples per ray re-drawn from the pdf of the marched samples' own weights
    smp.importance_sampling = True
    out2 = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
    r_info, r_s, r_e = O.ray_resampling(info, ts, te, w, 16)
    assert torch.equal(smp.packed_info.cpu(), r_info) and set(smp.packed_counts.cpu().tolist()) <= {0, 16}
    # (the pdf comes from the product's alphas, which differ from the oracle's by round-off: edges to 1e-4 of the march step count)
    assert_close("resampled starts", out2["ray_samples"].flat_starts.cpu(), r_s, rtol=0, atol=2e-4)
    assert_close("resampled ends", out2["ray_samples"].flat_ends.cpu(), r_e, rtol=0, atol=2e-4)
    assert torch.isfinite(out2["rgb"]).all() and float(out2["accumulation"].max()) <= 1.0 + 1e-4


# ------------------------------------------------------------------------------------------------ "grid" background field (config 5)
@pytest.mark.parametrize("features,smooth", [(2, False), (8, False), (2, True)])
def test_standalone_hash_grid_encode(device, features, smooth):
    """sdfhip_grid_encode_forward / _backward (the tcnn.Encoding("HashGrid") operator outside the fused fields) against
    oracle/hashgrid.py: features and the table gradient, positions inside and outside [0,1]^3 (wrapping)."""
    from oracle import hashgrid
    from sdfstudio_amd import _lib
    from sdfstudio_amd.fields.nerfacto_field import hash_grid_encode

    L, log2, base, max_res = 6, 11, 4, 96
    growth = math.exp((math.log(max_res) - math.log(base)) / (L - 1))
    lv = hashgrid.make_levels(L, features, log2, base, growth, smooth)
    cfg = _lib.GridCfg(L, features, log2, base, growth, 1 if smooth else 0)
    gen = torch.Generator().manual_seed(9)
    table = (torch.rand(lv.n_entries * features, generator=gen) * 2 - 1) * 0.5
    x = torch.rand(999, 3, generator=gen)
    x[:40] = x[:40] * 3.0 - 1.0  # outside the unit cube
    co = torch.randn(999, L * features, generator=gen)
    t_ref = table.clone().requires_grad_(True)
    f_ref = hashgrid.grid_encode(x, t_ref.view(lv.n_entries, features), lv)
    (f_ref * co).sum().backward()
    t = table.clone().to(device).requires_grad_(True)
    f = hash_grid_encode(t, x.to(device), cfg)
    (f * co.to(device)).sum().backward()
    assert_close("features", f, f_ref, rtol=1e-5, atol=1e-6)
    assert_close("table gradient", t.grad, t_ref.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("contraction", ["inf", None])
def test_nerf_background_field_fwd_bwd(device, contraction):
    """NeRFField, the reference's default background model (fields/vanilla_nerf_field.py:37-114; 8 x 256 ReLU MLP with a skip + 2 x 128
    head), on the fused sdfhip kernels (csrc/inst_d.hip) against the oracle's restatement (pinned on the reference's own class by
    test_oracle_nerf_background_field_against_reference) on IDENTICAL samples: density, rgb and every parameter gradient, with the
    element-wise gate.  n * s = 161 points: padded tail of the 128-point tiles."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames
    from sdfstudio_amd.fields.vanilla_nerf_field import NeRFField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    torch.manual_seed(13)
    from sdfstudio_amd.fields.vanilla_nerf_field import NeRFEncoding as _Enc

    fld = NeRFField(position_encoding=_Enc(in_dim=3, num_frequencies=10, min_freq_exp=0.0, max_freq_exp=9.0, include_input=True),
                    direction_encoding=_Enc(in_dim=3, num_frequencies=4, min_freq_exp=0.0, max_freq_exp=3.0, include_input=True),
                    spatial_distortion=SceneContraction(order=float("inf")) if contraction else None)
    with torch.no_grad():  # biases away from zero so that every ReLU pattern occurs
        for prm in fld.parameters():
            if prm.dim() == 1:
                prm.add_(0.1 * torch.randn_like(prm))
    fld = fld.to(device).train()
    n, s = 23, 7
    o, d, cam = O.synthetic_rays(n, seed=4)
    starts = torch.sort(torch.rand(n, s) * (6.0 if contraction else 1.5) + 0.3, dim=-1).values
    ends = starts + torch.rand(n, s) * 0.4 + 0.01
    rs = _bundle(o, d, cam, 0.3, 7.0, device).get_ray_samples(starts.to(device), ends.to(device))
    out = fld(rs)
    co = [torch.randn(n, s), torch.randn(n, s, 3)]
    (out[FieldHeadNames.DENSITY][..., 0] * co[0].to(device)).sum().add((out[FieldHeadNames.RGB] * co[1].to(device)).sum()).backward()

    def oracle(dtype):
        p = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in fld.state_dict().items()}
        ref = O.nerf_field(o.to(dtype), d.to(dtype), starts.to(dtype), ends.to(dtype), p, "", contraction)
        ((ref["density"] * co[0].to(dtype)).sum() + (ref["rgb"] * co[1].to(dtype)).sum()).backward()
        return ref, p

    ref, p = oracle(torch.float32)
    _, p64 = oracle(torch.float64)
    assert out[FieldHeadNames.DENSITY].shape == (n, s, 1) and out[FieldHeadNames.RGB].shape == (n, s, 3)
    assert_close("density", out[FieldHeadNames.DENSITY][..., 0], ref["density"], rtol=1e-4, atol=1e-6)
    assert_close("rgb", out[FieldHeadNames.RGB], ref["rgb"], rtol=1e-4, atol=1e-6)
    checked = 0
    for k, prm in fld.named_parameters():
        assert prm.grad is not None, k
        # 10-frequency encodings (2^9 x) amplify position round-off: the fp32 oracle itself is ~1e-4 from its fp64 evaluation on the
        # first layer; the bar is the fp32 round-off class of the reference path, or 1e-3 of the maximum
        assert_fp32_class(f"grad {k}", prm.grad, p[k].grad, p64[k].grad, factor=3.0, atol=1e-3 * p64[k].grad.abs().max().item())
        checked += 1
    assert checked == 2 * (8 + 2 + 1 + 1)


@pytest.mark.parametrize("training", [True, False])
def test_nerfacto_background_field_fwd_bwd(device, training):
    """TCNNNerfactoField mirror (fields/nerfacto_field.py:65-332, the "grid" background of BASELINE config 5) against the oracle's
    restatement (pinned on the reference's own class by test_oracle_nerfacto_background_field_against_reference): density, rgb and
    every parameter gradient."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.models.neus_facto import SceneContraction
    from oracle import hashgrid

    torch.manual_seed(11)
    fld = TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=9, num_levels=5, max_res=48, log2_hashmap_size=9,
                            spatial_distortion=SceneContraction(order=float("inf")))
    with torch.no_grad():
        fld.mlp_base.table.copy_((torch.rand_like(fld.mlp_base.table) * 2 - 1) * 0.4)
    fld = fld.to(device).train(training)
    n, s = 23, 7
    o, d, cam = O.synthetic_rays(n, seed=4)
    cam = cam % 9
    starts = torch.sort(torch.rand(n, s) * 6.0 + 0.3, dim=-1).values
    ends = starts + torch.rand(n, s) * 0.4 + 0.01
    rs = _bundle(o, d, cam, 0.3, 7.0, device).get_ray_samples(starts.to(device), ends.to(device))
    out = fld(rs)
    co = [torch.randn(n, s), torch.randn(n, s, 3)]
    (out[FieldHeadNames.DENSITY][..., 0] * co[0].to(device)).sum().add((out[FieldHeadNames.RGB] * co[1].to(device)).sum()).backward()
    names = {"bg.mlp_base.table": "mlp_base.table", "bg.mlp_base.w1": "mlp_base.w1", "bg.mlp_base.w2": "mlp_base.w2",
             "bg.mlp_head.w1": "mlp_head.w1", "bg.mlp_head.w2": "mlp_head.w2", "bg.mlp_head.w3": "mlp_head.w3",
             "bg.embedding_appearance.embedding.weight": "embedding_appearance.embedding.weight"}
    sd = {k: v.detach().cpu() for k, v in fld.state_dict().items()}
    p = {k: sd[v].clone().requires_grad_(True) for k, v in names.items()}
    growth = math.exp((math.log(48) - math.log(16)) / 4)
    lv = hashgrid.make_levels(5, 2, 9, 16, growth, False)
    ref = O.nerfacto_field(o, d, starts, ends, cam, p, "bg.", lv, training=training)
    ((ref["density"] * co[0]).sum() + (ref["rgb"] * co[1]).sum()).backward()
    assert_close("density", out[FieldHeadNames.DENSITY][..., 0], ref["density"], rtol=1e-4, atol=1e-6)
    assert_close("rgb", out[FieldHeadNames.RGB], ref["rgb"], rtol=1e-4, atol=1e-6)
    grads = dict(fld.named_parameters())
    for k, v in names.items():
        if p[k].grad is None or (not training and "embedding" in k):
            continue
        assert_close(f"grad {v}", grads[v].grad, p[k].grad, rtol=1e-3, atol=1e-8)


@pytest.mark.parametrize("S,white", [(24, False), (96, True), (130, False)])
def test_volsdf_render_fwd_bwd(device, S, white):
    """renderers.volsdf_render (models/volsdf.py:62-79 in one launch: Laplace density -> weights -> rgb / depth / normal /
    accumulation + the transmittance in front of the last sample) against the oracle's per-statement composition, forward and
    backward w.r.t. sdf, gradients, rgb and beta, with every output carrying a cotangent."""
    from sdfstudio_amd.model_components.renderers import volsdf_render

    gen = torch.Generator().manual_seed(S)
    n = 37
    sdf = torch.randn(n, S, generator=gen) * 0.3
    sdf[:, 0] = 0.0  # sign(0) = 0 branch
    grad = torch.randn(n, S, 3, generator=gen)
    rgb = torch.rand(n, S, 3, generator=gen)
    starts = torch.sort(torch.rand(n, S, generator=gen) * 4 + 0.5, dim=-1).values
    ends = torch.cat([starts[:, 1:], starts[:, -1:] + 0.05], dim=-1)
    beta = torch.tensor([0.07])
    bg = torch.ones(3) if white else None
    co = [torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen), torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen),
          torch.randn(n, S, generator=gen), torch.randn(n, generator=gen)]

    def run(f, dev, dtype):
        leaves = [t.clone().to(dev, dtype).requires_grad_(True) for t in (sdf, grad, rgb, beta)]
        outs = f(*leaves, starts.to(dev, dtype), ends.to(dev, dtype), None if bg is None else bg.to(dev, dtype))
        loss = sum((o * c.to(dev, dtype)).sum() for o, c in zip(outs, co))
        loss.backward()
        return [o.detach() for o in outs], [l.grad for l in leaves]

    def oracle(sdf_, grad_, rgb_, beta_, st, en, bg_):
        density = O.laplace_density(sdf_, beta_)
        w, trans = O.weights_and_transmittance_from_density(density, en - st)
        out_rgb, depth, normal, acc = O.render(w, rgb_, F.normalize(grad_, p=2, dim=-1), st, en, bg_)
        return out_rgb, depth, normal, acc, w, trans[:, -1]

    def product(sdf_, grad_, rgb_, beta_, st, en, bg_):
        out_rgb, depth, normal, acc, w, density, bgt = volsdf_render(sdf_, grad_, rgb_, beta_, st, en, bg_)
        return out_rgb, depth, normal, acc, w, bgt

    got, ggot = run(product, device, torch.float32)
    ref, gref = run(oracle, "cpu", torch.float64)
    for name, a, b in zip(["rgb", "depth", "normal", "accumulation", "weights", "bg transmittance"], got, ref):
        assert_close(name, a, b.float(), rtol=1e-4, atol=1e-6)
    for name, a, b in zip(["d / d sdf", "d / d gradient", "d / d rgb", "d / d beta"], ggot, gref):
        assert_close(name, a, b.float(), rtol=1e-3, atol=1e-6)


def test_packed_resampling_against_oracle(device):
    """sdfhip_packed_resample (nerfacc.ray_resampling as ray_samplers.py:1496-1498 calls it: NeuSAccSampler(importance_sampling=True))
    against the oracle's fp32 restatement, bit for bit: rays with 0, 1 and many samples, zero weights (padded pdf), and the sampler's
    own use of it after a grid update."""
    from sdfstudio_amd.model_components.ray_samplers import resample_packed

    gen = torch.Generator().manual_seed(6)
    counts = torch.tensor([0, 3, 40, 1, 0, 17, 200, 2])
    offs = torch.cumsum(counts, 0) - counts
    info = torch.stack([offs, counts], -1)
    st, en = [], []
    for c in counts.tolist():
        e = torch.sort(torch.rand(c + 1, generator=gen) * 3 + 0.5)[0]
        st.append(e[:-1])
        en.append(e[1:])
    st, en = torch.cat(st)[:, None], torch.cat(en)[:, None]
    w = torch.rand(int(counts.sum()), generator=gen) * 0.2
    w[int(offs[3])] = 0.0          # a ray whose only sample has no weight
    w[int(offs[5]):int(offs[5]) + 17] = 0.0  # a ray of zero weights: uniform through the padding
    ref_info, ref_s, ref_e = O.ray_resampling(info, st, en, w, 16)
    g_info, g_counts, g_ri, g_s, g_e = resample_packed(info.to(device), counts.to(torch.int32).to(device), st.to(device), en.to(device),
                                                       w.to(device), 16)
    assert torch.equal(g_info.cpu(), ref_info) and torch.equal(g_counts.cpu().long(), ref_info[:, 1])
    assert torch.equal(g_ri.cpu(), torch.repeat_interleave(torch.arange(len(counts)), ref_info[:, 1]))
    assert torch.equal(g_s.cpu(), ref_s) and torch.equal(g_e.cpu(), ref_e)
    assert bool((ref_e >= ref_s).all())
