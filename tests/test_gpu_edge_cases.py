"""Edge cases of the hot path through the product's surface (-m gpu): batch sizes the tile shapes do not divide (1 ray, 1 sample, 33 rays),
a ray's results independent of the rays beside it, degenerate intervals (near == far), empty inputs at the native boundary, and the
same for the mesh library (the smallest legal volume, a volume without a surface, a surface that touches every face of the lattice).

What the reference does in these cases is plain torch semantics on [N, S] tensors (empty tensors flow through; every per-ray result is a
function of that ray alone, except the expected-depth clip to the batch's global [min, max] of sample positions, renderers.py:257);
the oracle restates exactly that, so the checks below are oracle comparisons where the oracle has the function and per-ray independence
(bit for bit: a point's dot products are accumulated in the same order whichever tile it sits in) elsewhere.
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, load_golden, product_model_from_params, small_oracle_cfg
from oracle import sdf_path as O

pytestmark = pytest.mark.gpu


def _bundle(o, d, cam, near, far, device):
    from sdfstudio_amd.cameras.rays import RayBundle

    n = o.shape[0]
    return RayBundle(origins=o.to(device), directions=d.to(device), pixel_area=torch.ones(n, 1, device=device),
                     directions_norm=torch.ones(n, 1, device=device), camera_indices=cam[:, None].to(device),
                     nears=torch.full((n, 1), near, device=device), fars=torch.full((n, 1), far, device=device))


def _eval_model(device):
    g = load_golden("eval")
    cfg = small_oracle_cfg()
    return product_model_from_params(g["param"], cfg, device).eval(), cfg


PER_RAY = ("rgb", "accumulation", "normal", "weights")


@pytest.mark.parametrize("n", [1, 2, 31, 33, 129])
def test_a_rays_results_do_not_depend_on_the_batch_it_sits_in(device, n):
    """Eval mode (deterministic sampling): the first n rays of a 160-ray batch, rendered alone, give the same rgb / accumulation / normal /
    weights bit for bit - n = 1, 2 (a fraction of one 32-point tile), 31 / 33 / 129 (tile and workgroup boundaries)."""
    model, cfg = _eval_model(device)
    o, d, cam = O.synthetic_rays(160, seed=5)
    with torch.no_grad():
        full = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
        part = model(_bundle(o[:n], d[:n], cam[:n], cfg.near, cfg.far, device))
    for k in PER_RAY:
        a, b = part[k], full[k][:n]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert torch.isfinite(a).all(), k
        assert torch.equal(a, b), (k, float((a - b).abs().max()))
    # expected depth is clipped to the BATCH's [min, max] sample position (renderers.py:257): equal wherever the clip is not active
    da, db = part["depth"][..., 0], full["depth"][:n, 0]
    inner = (db > 0.55) & (db < 4.4)
    assert torch.equal(da[inner], db[inner])


def test_one_sample_per_ray_and_one_ray(device):
    """S = 1 through the spaced sampler, the field, the NeuS alpha / weights and the renderers against the oracle's per-ray functions:
    one interval [near, far], weight = alpha, T = 1."""
    from sdfstudio_amd.model_components.ray_samplers import UniformSampler
    from sdfstudio_amd.model_components.renderers import AccumulationRenderer, RGBRenderer

    model, cfg = _eval_model(device)
    for n in (1, 5):
        o, d, cam = O.synthetic_rays(n, seed=2)
        rb = _bundle(o, d, cam, 1.0, 3.5, device)
        rs = UniformSampler(single_jitter=True).eval()(rb, num_samples=1)
        assert rs.frustums.starts.shape == (n, 1, 1)
        assert_close("starts", rs.flat_starts[:, 0], torch.full((n,), 1.0), rtol=0, atol=1e-6)
        assert_close("ends", rs.flat_ends[:, 0], torch.full((n,), 3.5), rtol=0, atol=1e-6)
        with torch.no_grad():
            fo = model.field(rs, return_alphas=True)
        from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

        alpha = fo[H.ALPHA]
        assert alpha.shape == (n, 1, 1) and torch.isfinite(alpha).all() and (alpha >= 0).all() and (alpha <= 1).all()
        w, T = rs.get_weights_and_transmittance_from_alphas(alpha)
        assert torch.allclose(w, alpha, atol=1e-7)  # T_0 = 1 (up to the 1e-7 of rays.py:222's cumprod argument)
        acc = AccumulationRenderer()(w)
        assert torch.equal(acc, w.sum(-2))
        rgb = RGBRenderer(background_color=torch.ones(3)).train()(fo[H.RGB], w)
        want = (w * fo[H.RGB]).sum(-2) + (1 - acc)
        assert_close("rgb", rgb, want.cpu(), rtol=0, atol=1e-6)


def test_zero_length_rays_render_the_background(device):
    """near == far: every interval has zero length; the NeuS alpha of an empty interval is 1e-5 / (cdf + 1e-5) (sdf_field.py:519-521: the two
    1e-5 terms), i.e. 1e-5 at these points outside the surface (cdf = 1) - not 0 - so each of the 16 samples weighs <= 1e-5, rgb is the
    background to 2e-4 and nothing is NaN (the expected depth divides by accumulation + 1e-10, renderers.py:252); rays of normal length in
    the same batch are not disturbed."""
    model, cfg = _eval_model(device)
    o, d, cam = O.synthetic_rays(64, seed=7)
    rb = _bundle(o, d, cam, cfg.near, cfg.far, device)
    ref = None
    with torch.no_grad():
        ref = model(_bundle(o, d, cam, cfg.near, cfg.far, device))
        rb.fars[:16] = rb.nears[:16]
        out = model(rb)
    for k in PER_RAY + ("depth",):
        assert torch.isfinite(out[k]).all(), k
    w = out["weights"][:16]
    assert float(w.min()) > 0.99e-5 * (1 - 1e-5) ** 16 and float(w.max()) <= 1.01e-5  # cdf = sigmoid(sdf inv_s) is within 1e-3 of 1 out here
    assert float((out["accumulation"][:16] - w.sum(-2)).abs().max()) < 1e-9
    bg = out["rgb"][:16]
    assert float((bg - bg[:1]).abs().max()) < 2e-4  # one background colour on all of them, up to 16 x 1e-5 of the samples' own colours
    for k in PER_RAY:
        assert torch.equal(out[k][16:], ref[k][16:]), k


def test_empty_inputs_at_the_native_boundary(device):
    """n = 0 through the entry points a caller can reach with an empty chunk (an eval image whose last chunk is empty, a NeuS-acc step in
    which the march keeps no sample): the call returns without a launch error and the outputs have the empty shapes torch would give
    (an EMPTY tensor's data_ptr() is 0: _lib.ptr hands the entry points a valid address instead of a NULL they would refuse)."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H
    from sdfstudio_amd.model_components.ray_samplers import PDFSampler, UniformLinDispPiecewiseSampler, UniformSampler
    from sdfstudio_amd.model_components.renderers import AccumulationRenderer, RGBRenderer

    model, cfg = _eval_model(device)
    z3, z1 = torch.zeros(0, 3), torch.zeros(0, dtype=torch.long)
    rb = _bundle(z3, z3, z1, 0.5, 4.5, device)
    errors = []

    def attempt(name, fn):
        try:
            out = fn()
            torch.cuda.synchronize()
            return out
        except Exception as e:  # noqa: BLE001 - every failure is collected and reported together
            errors.append(f"{name}: {type(e).__name__}: {str(e)[:160]}")
            return None

    rs = attempt("UniformSampler", lambda: UniformSampler(single_jitter=True).eval()(rb, num_samples=8))
    attempt("UniformLinDispPiecewiseSampler", lambda: UniformLinDispPiecewiseSampler(single_jitter=True).train()(rb, num_samples=8))
    if rs is not None:
        assert rs.frustums.starts.shape == (0, 8, 1)
        rs2 = attempt("PDFSampler", lambda: PDFSampler(single_jitter=True, include_original=False).eval()(
            rb, rs, torch.zeros(0, 8, 1, device=device), num_samples=4))
        assert rs2 is None or rs2.frustums.starts.shape == (0, 4, 1)
        with torch.no_grad():
            sdf = attempt("get_sdf", lambda: model.field.get_sdf(rs))
            assert sdf is None or sdf.shape == (0, 8, 1)
            geo = attempt("forward_geonetwork", lambda: model.field.forward_geonetwork(torch.zeros(0, 3, device=device)))
            assert geo is None or (geo.shape[0] == 0 and geo.shape[1] == 1 + model.field.config.geo_feat_dim)
            fo = attempt("field.forward (no_grad)", lambda: model.field(rs, return_alphas=True))
        if fo is not None:
            assert fo[H.RGB].shape == (0, 8, 3) and fo[H.ALPHA].shape == (0, 8, 1)
            w, _ = rs.get_weights_and_transmittance_from_alphas(fo[H.ALPHA])
            assert w.shape == (0, 8, 1)
            assert AccumulationRenderer()(w).shape == (0, 1)
            assert RGBRenderer(background_color=torch.zeros(3))(fo[H.RGB], w).shape == (0, 3)
        dens = attempt("proposal density", lambda: model.proposal_networks[0].density_fn(rs))
        assert dens is None or dens.shape == (0, 8, 1)
        wts = attempt("get_weights (density)", lambda: rs.get_weights(torch.zeros(0, 8, 1, device=device)))
        assert wts is None or wts.shape == (0, 8, 1)
        # with a graph: an empty field evaluation contributes zero gradients, not an error
        model.train()

        def train_empty():
            fo = model.field(rs, return_alphas=True)
            (fo[H.RGB].sum() + fo[H.SDF].sum() + fo[H.GRADIENT].sum()).backward()
            return fo

        attempt("field.forward + backward (train)", train_empty)
        for name, p in model.field.named_parameters():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name  # (round 6: theta_bar came back uninitialised from the empty call)

        def prop_empty():
            net = model.proposal_networks[0]
            net.density_fn(rs).sum().backward()
            for name, p in net.named_parameters():
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name

        attempt("proposal density + backward (train)", prop_empty)
        model.eval()
    with torch.no_grad():
        out = attempt("model(rb) eval", lambda: model(rb))
        assert out is None or (out["rgb"].shape == (0, 3) and out["accumulation"].shape == (0, 1))
    assert not errors, "\n".join(errors)


def test_mesh_library_edge_volumes(device):
    """libsdfmesh.so on the smallest legal volume (2 x 2 x 2: one cell), a volume without a surface at the level inside its range check,
    and a surface cut by every face of the lattice (no neighbour for the normals' central differences) - against the oracle, bit for bit."""
    from oracle import marching_cubes as OM
    from sdfstudio_amd.utils import marching_cubes as MC

    rng = np.random.default_rng(3)
    cases = [rng.standard_normal((2, 2, 2)).astype(np.float32),
             rng.standard_normal((2, 3, 65)).astype(np.float32),      # one row longer than a 64-bit word of point bits
             rng.standard_normal((5, 4, 3)).astype(np.float32)]
    x = np.linspace(-1, 1, 9, dtype=np.float32)
    cases.append((x[:, None, None] + 0.5 * x[None, :, None] - 0.25 * x[None, None, :]).astype(np.float32))  # a plane through every face
    for vol in cases:
        want = OM.marching_cubes(vol, 0.0)
        got = MC.marching_cubes(torch.from_numpy(vol).to(device), 0.0)
        for name, a, b in zip(("verts", "faces", "normals", "values"), got, want):
            assert a.shape == tuple(b.shape), (vol.shape, name, a.shape, b.shape)
            assert np.array_equal(a.cpu().numpy(), b), (vol.shape, name)
    # inside the data range, but no cell straddles the level once the mask removes them: scikit-image's "No surface found"
    vol = np.ones((4, 4, 4), np.float32)
    vol[0, 0, 0] = -1.0
    mask = np.ones((4, 4, 4), bool)
    mask[:2, :2, :2] = False
    with pytest.raises(RuntimeError, match="No surface found"):
        MC.marching_cubes(torch.from_numpy(vol).to(device), 0.0, mask=torch.from_numpy(mask).to(device))
    with pytest.raises(ValueError, match="within volume data range"):
        MC.marching_cubes(torch.from_numpy(vol).to(device), 2.0)
