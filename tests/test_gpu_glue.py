"""Host-side glue that moved into native launches (round 4: config 5's per-step ATen launches): the pinhole ray generator, the geometry
network's ray entry (frustum positions + scene contraction inside the kernel) and the permuted parameter vector of the background field."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    assert torch.cuda.is_available(), "the -m gpu tests need a HIP device"
    return torch.device("cuda:0")


def test_generate_pinhole_rays_against_the_torch_statement(device):
    """sdfhip_generate_rays (PixelSampler + RayGenerator arithmetic, pixel_samplers.py:47-50 / cameras.py:462-640) against the per-statement
    torch version bench.py used until round 4: same uniforms -> same cameras and pixels (exact), unit directions and their norms to 1 ulp."""
    import bench

    centers, rot = bench.synthetic_cameras(device)
    for n, seed in ((4096, 3), (1, 5), (257, 7)):
        g1, g2 = torch.Generator(device=device), torch.Generator(device=device)
        g1.manual_seed(seed)
        g2.manual_seed(seed)
        o, d, norm, cam = bench.draw_rays(centers, rot, n, g1)
        o2, d2, norm2, cam2 = bench.draw_rays_torch(centers, rot, n, g2)
        assert torch.equal(cam, cam2) and cam.dtype == torch.int64
        assert torch.equal(o, o2)
        assert_close("directions", d, d2, rtol=0, atol=3e-7)
        assert_close("directions_norm", norm, norm2, rtol=3e-7, atol=0)
        assert_close("unit length", d.norm(dim=-1), torch.ones(n, device=device), rtol=0, atol=3e-7)
    # edge draws: u = 0 and u just below 1 (last camera, last pixel)
    from sdfstudio_amd.cameras.rays import generate_pinhole_rays

    u = torch.tensor([[0.0, 0.0, 0.0], [0.999999, 0.999999, 0.999999]], device=device)
    o, d, norm, cam = generate_pinhole_rays(u, centers, rot, 384, 384, 925.5, 922.6, 199.4, 198.1)
    assert cam.tolist() == [0, 48]
    exp = torch.tensor([[(0.5 - 199.4) / 925.5, (0.5 - 198.1) / 922.6, 1.0], [(383.5 - 199.4) / 925.5, (383.5 - 198.1) / 922.6, 1.0]], device=device)
    dw = torch.einsum("nij,nj->ni", rot[cam], exp)
    assert_close("corner pixel directions", d * norm, dw, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("order", [float("inf"), None])
def test_background_field_ray_entry_equals_the_host_composed_positions(device, order):
    """TCNNNerfactoField.get_density through sdfhip_geo_forward_rays (frustum mid points + SceneContraction inside the kernel) against the
    same field fed host-computed positions (frustums.get_positions() -> SceneContraction.forward -> sdfhip_geo_forward): density, colour
    and every parameter gradient.  Positions agree to an ulp, so the grid features do to ~1e-5 of their scale."""
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.fields.field_heads import FieldHeadNames
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    torch.manual_seed(5)
    fld = TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=7, num_levels=6, max_res=64, log2_hashmap_size=10,
                            spatial_distortion=SceneContraction(order=order))
    with torch.no_grad():
        fld.mlp_base.table.copy_((torch.rand_like(fld.mlp_base.table) * 2 - 1) * 0.4)
    fld = fld.to(device).train()
    n, s = 61, 9
    o = torch.randn(n, 3, device=device) * 0.4
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=device), dim=-1)
    starts = torch.sort(torch.rand(n, s, device=device) * 7.0, dim=-1).values  # inside and far outside the unit ball
    ends = starts + torch.rand(n, s, device=device) * 0.5 + 0.01
    cam = torch.randint(0, 7, (n, 1), device=device)
    co = [torch.randn(n, s, device=device), torch.randn(n, s, 3, device=device)]

    def run(native_positions):
        fld.zero_grad(set_to_none=True)
        fld._contract = (1 if order == float("inf") else 2) if native_positions else 0  # 0: the host-composed path
        rs = RayBundle(origins=o, directions=d, camera_indices=cam).get_ray_samples(starts, ends)
        out = fld(rs)
        ((out[FieldHeadNames.DENSITY][..., 0] * co[0]).sum() + (out[FieldHeadNames.RGB] * co[1]).sum()).backward()
        return out, {k: p.grad.clone() for k, p in fld.named_parameters() if p.grad is not None}

    out_n, g_n = run(True)
    out_h, g_h = run(False)
    assert_close("density", out_n[FieldHeadNames.DENSITY], out_h[FieldHeadNames.DENSITY], rtol=2e-4, atol=1e-6)
    assert_close("rgb", out_n[FieldHeadNames.RGB], out_h[FieldHeadNames.RGB], rtol=0, atol=2e-5)
    assert set(g_n) == set(g_h) and len(g_n) >= 7
    for k in g_n:
        assert_close(f"grad {k}", g_n[k], g_h[k], rtol=2e-3, atol=1e-7 + 2e-4 * float(g_h[k].abs().max()))


def test_permuted_theta_writes_gradient_slots(device):
    """The background field's parameter vector (_PermutedTheta): same forward as cat + index_select, and under a flat gradient buffer the
    five weight gradients land in their slots without autograd's index_add_ / split / copy chain - equal to autograd's own result."""
    from sdfstudio_amd.distributed import FlatGradients
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    torch.manual_seed(2)
    fld = TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=3, num_levels=4, max_res=32, log2_hashmap_size=8,
                            spatial_distortion=SceneContraction(order=float("inf"))).to(device).train()
    params = [fld.mlp_base.w1, fld.mlp_base.w2, fld.mlp_head.w1, fld.mlp_head.w2, fld.mlp_head.w3]
    theta = fld._theta()
    assert fld._theta_invs is not None and theta.grad_fn is not None and "PermutedTheta" in type(theta.grad_fn).__name__
    w = torch.randn_like(theta)
    ref = torch.cat([p.reshape(-1) for p in params] + [fld._theta_zero]).index_select(0, fld._theta_src)
    assert torch.equal(theta.detach(), ref.detach())
    want = torch.autograd.grad((ref * w).sum(), params)
    flat = FlatGradients(list(fld.parameters()))
    loss = (theta * w).sum()
    flat.zero(loss)
    loss.backward()
    flat.finish()
    for p, g in zip(params, want):
        assert torch.equal(p.grad, g)
        assert p.grad.data_ptr() >= flat.flat.data_ptr() and p.grad.data_ptr() < flat.flat.data_ptr() + flat.flat.numel() * 4


def _ray_vectors():
    import os

    import numpy as np

    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays_reference.npz")))


def test_generate_rays_against_the_references_cameras(device):
    """sdfhip_generate_rays against vectors minted by the REFERENCE's own data path (tests/golden/make_golden_rays.py: the PixelSampler's
    floor(rand * [C, H, W]), image_coords + 0.5 and Cameras.generate_rays for perspective cameras): same cameras and origins exactly, unit
    directions and directions_norm to an ulp."""
    from sdfstudio_amd.cameras.rays import generate_pinhole_rays

    g = _ray_vectors()
    fx, fy, cx, cy, H, W, C = g["rays/intrinsics"]
    t = lambda k: torch.tensor(g[k]).to(device)  # noqa: E731
    o, d, norm, cam = generate_pinhole_rays(t("rays/u"), t("rays/centers"), t("rays/rot"), int(H), int(W), fx, fy, cx, cy)
    assert torch.equal(cam.cpu(), torch.tensor(g["rays/indices"][:, 0]))
    assert torch.equal(o, t("rays/origins"))
    assert_close("directions", d, t("rays/directions"), rtol=0, atol=3e-7)
    assert_close("directions_norm", norm, t("rays/directions_norm"), rtol=3e-7, atol=0)


def test_whole_image_rays_against_the_references_cameras(device):
    """generate_image_rays (every pixel of one camera, batch shape [H, W]: the eval path's input) against the reference's own
    Cameras.generate_rays(camera_indices=i) and RayBundle.get_row_major_sliced_ray_bundle (tests/golden/make_golden_rays.py, image/*)."""
    from sdfstudio_amd.cameras.rays import generate_image_rays

    g = _ray_vectors()
    fx, fy, cx, cy, H, W, C = g["rays/intrinsics"]
    ci = int(g["image/camera"])
    t = lambda k: torch.tensor(g[k]).to(device)  # noqa: E731
    rb = generate_image_rays(t("rays/centers")[ci], t("rays/rot")[ci], int(H), int(W), fx, fy, cx, cy, camera_index=ci)
    assert tuple(rb.origins.shape) == (int(H), int(W), 3) and len(rb) == int(H) * int(W)
    assert torch.equal(rb.origins, t("image/origins")) and torch.equal(rb.camera_indices, t("image/camera_indices"))
    assert_close("image directions", rb.directions, t("image/directions"), rtol=0, atol=3e-7)
    assert_close("image directions_norm", rb.directions_norm, t("image/directions_norm"), rtol=3e-7, atol=0)
    ch = rb.get_row_major_sliced_ray_bundle(100, 164)
    assert tuple(ch.origins.shape) == (64, 3) and torch.equal(ch.camera_indices, t("image/chunk_100_164_camera_indices"))
    assert_close("chunk directions", ch.directions, t("image/chunk_100_164_directions"), rtol=0, atol=3e-7)


def test_get_outputs_for_camera_ray_bundle_is_the_chunked_forward(device):
    """models/base_model.py:165-189: an image's rays in row-major chunks of eval_num_rays_per_chunk (here 100: 24 x 20 = 480 rays = 4 chunks
    and a ragged fifth of 80) give, per pixel, what ONE forward over all of them gives - bit for bit for every per-ray output (the expected
    depth's clip to the batch's sample range, renderers.py:257, is per chunk in the reference as well: compared where it is not active) -
    viewed [H, W, -1]; list outputs are dropped; no graph is built."""
    from helpers import load_golden, product_model_from_params, small_oracle_cfg
    from sdfstudio_amd.cameras.rays import generate_image_rays

    g = load_golden("eval")
    cfg = small_oracle_cfg()
    model = product_model_from_params(g["param"], cfg, device).eval()
    model.config.eval_num_rays_per_chunk = 100
    H, W = 24, 20
    center = torch.tensor([0.3, -2.4, 1.2], device=device)
    z = -center / center.norm()
    x = torch.linalg.cross(z, torch.tensor([0.0, 0.0, 1.0], device=device))
    x = x / x.norm()
    rot = torch.stack([x, torch.linalg.cross(z, x), z], dim=1)  # columns: x right, y down, z forward
    rb = generate_image_rays(center, rot, H, W, 30.0, 30.0, W / 2, H / 2)
    out = model.get_outputs_for_camera_ray_bundle(rb)
    with torch.no_grad():
        whole = model(rb.flatten())
    assert tuple(out["rgb"].shape) == (H, W, 3) and tuple(out["accumulation"].shape) == (H, W, 1) and tuple(out["normal"].shape) == (H, W, 3)
    assert not out["rgb"].requires_grad and "weights_list" not in out and "ray_samples_list" not in out
    assert float(out["accumulation"].max()) > 0.5, "the camera must see the surface"
    for k in ("rgb", "accumulation", "normal", "normal_vis"):
        assert torch.equal(out[k].reshape(H * W, -1), whole[k].reshape(H * W, -1)), k
    d0, d1 = out["depth"].reshape(-1), whole["depth"].reshape(-1)
    inner = (d1 > cfg.near * 1.1) & (d1 < cfg.far * 0.97)
    assert torch.equal(d0[inner], d1[inner])


@pytest.mark.parametrize("name,order", [("inf", float("inf")), ("l2", None)])
def test_ray_entry_positions_against_the_references_frustums_and_contraction(device, name, order):
    """The positions sdfhip_geo_forward_rays forms inside the encode kernel (x_out) against the reference's Frustums.get_positions() /
    get_start_positions() followed by its SceneContraction (vectors of tests/golden/make_golden_rays.py), both norms, mid and start points."""
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.fields.sdf_field import _GeoNetRaysFunction
    from sdfstudio_amd.models.neus_facto import SceneContraction

    g = _ray_vectors()
    fld = TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=3, num_levels=4, max_res=32, log2_hashmap_size=8,
                            spatial_distortion=SceneContraction(order=order)).to(device)
    t = lambda k: torch.tensor(g[k]).to(device)  # noqa: E731
    o, d, st, en = t("pos/origins"), t("pos/directions"), t("pos/starts"), t("pos/ends")
    mask = torch.ones(fld.grid_cfg.n_levels * fld.grid_cfg.n_features, device=device)
    with torch.no_grad():
        theta = fld._theta()
        for ends, key in ((en, f"pos/mid_{name}"), (None, f"pos/start_{name}")):
            _, _, x = _GeoNetRaysFunction.apply(theta, fld.mlp_base.table, fld._native, o, d, st, ends, mask)
            assert_close(key, x.view(*st.shape, 3), t(key), rtol=0, atol=4e-7)


def test_field_methods_of_the_plugin_surface(device):
    """The SDFField methods a reference caller may use besides forward (fields/sdf_field.py, fields/base_field.py):
    get_colors on caller-supplied tensors reproduces the colour the fused forward computed from the same inputs; density_fn equals
    get_density's density at zero-length frustums; get_normals raises the reference's assertion."""
    from helpers import load_golden, product_model_from_params, small_oracle_cfg
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.fields.field_heads import FieldHeadNames
    from oracle import sdf_path as O

    g = load_golden("train")
    model = product_model_from_params(g["param"], small_oracle_cfg(), device).train()
    f = model.field
    n, s = 32, 8
    o, d, cam = O.synthetic_rays(n, seed=5)
    starts = torch.sort(torch.rand(n, s) * 2.0 + 1.5, dim=-1)[0]
    rb = RayBundle(origins=o.to(device), directions=d.to(device), camera_indices=cam[:, None].to(device),
                   nears=torch.full((n, 1), 0.5, device=device), fars=torch.full((n, 1), 4.5, device=device))
    rs = rb.get_ray_samples(starts.to(device), starts.to(device) + 0.1)
    out = f(rs)
    x = f.spatial_distortion(rs.frustums.get_start_positions()) if f.spatial_distortion is not None else rs.frustums.get_start_positions()
    feat = f.forward_geonetwork(x.reshape(-1, 3))[:, 1:].reshape(n, s, -1)  # (get_outputs evaluates the geometry network at the contracted points)
    rgb = f.get_colors(x, rs.frustums.directions, out[FieldHeadNames.GRADIENT], feat, rs.camera_indices.reshape(n, 1).expand(n, s))
    assert rgb.shape == (n, s, 3)
    assert_close("get_colors vs the fused forward", rgb, out[FieldHeadNames.RGB], rtol=0, atol=5e-5)
    # the gradient of the colour reaches the colour network's parameters and the supplied normal
    gin = out[FieldHeadNames.GRADIENT].detach().clone().requires_grad_(True)
    f.get_colors(x, rs.frustums.directions, gin, feat.detach(), rs.camera_indices.reshape(n, 1).expand(n, s)).sum().backward()
    assert gin.grad is not None and float(gin.grad.abs().max()) > 0.0 and float(f.clin0.weight_v.grad.abs().max()) > 0.0
    pos = rs.frustums.get_start_positions()
    with torch.no_grad():
        dens, _ = f.get_density(rs)
        assert torch.equal(f.density_fn(pos), dens)
    with pytest.raises(AssertionError):
        f.get_normals()


def test_spaced_sampler_from_spacing_functions(device):
    """SpacedSampler(spacing_fn, spacing_fn_inv, ...) as the reference constructs it (ray_samplers.py:66-78) equals the named sampler."""
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.model_components.ray_samplers import LinearDisparitySampler, SpacedSampler

    n = 10
    rb = RayBundle(origins=torch.zeros(n, 3, device=device), directions=torch.ones(n, 3, device=device), pixel_area=torch.ones(n, 1, device=device),
                   nears=torch.full((n, 1), 2.0, device=device), fars=torch.full((n, 1), 4.0, device=device))
    a = SpacedSampler(spacing_fn=lambda x: 1 / x, spacing_fn_inv=lambda x: 1 / x, num_samples=15).eval()(rb)
    b = LinearDisparitySampler(num_samples=15).eval()(rb)
    assert a.frustums.get_positions().shape[-2] == 15  # the reference's own check (tests/model_components/test_ray_sampler.py)
    assert torch.equal(a.frustums.starts, b.frustums.starts) and torch.equal(a.frustums.ends, b.frustums.ends)


@pytest.mark.parametrize("name", ["neus-facto", "neus", "mono-neus", "volsdf", "monosdf", "unisurf", "neus-acc"])
def test_method_presets_train_a_step(device, name):
    """configs/method_configs.py end to end: the preset's model config builds its model, the preset's optimizer dictionary (the reference's
    config objects) builds the fused optimiser, the training callbacks run as the reference's trainer runs them, one step trains: finite
    losses under the reference's keys, parameters move."""
    import copy

    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.configs.method_configs import method_configs
    from sdfstudio_amd.engine.callbacks import TrainingCallbackLocation
    from sdfstudio_amd.engine.optimizers import Optimizers
    from sdfstudio_amd.models.neus_facto import SceneBox
    import bench as B

    m = method_configs[name]
    cfg = copy.deepcopy(m.model)
    cfg.background_model = "none"  # (the presets' default NeRFField background: its own tests; here the method's own path)
    torch.manual_seed(0)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5)
    model = cfg.setup(scene_box=box, num_train_data=49).to(device).train()
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    opts = Optimizers({k: m.optimizers[k] for k in groups}, groups)
    callbacks = model.get_training_callbacks(None)
    gen = torch.Generator(device=device)
    gen.manual_seed(1)
    centers, rot = B.synthetic_cameras(device)
    n = 256
    before = {k: p.detach().clone() for k, p in list(model.field.named_parameters())[:4]}
    for step in range(2):
        for cb in callbacks:
            cb.run_callback_at_location(step, TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        o, d, norm, cam = B.draw_rays(centers, rot, n, gen)
        out = model(RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None]))
        batch = {"image": torch.rand(n, 3, device=device, generator=gen)}
        if "mono" in name:
            batch["depth"] = torch.rand(n, device=device, generator=gen)
            batch["normal"] = torch.nn.functional.normalize(torch.randn(n, 3, device=device, generator=gen), dim=-1)
        loss = model.get_loss_dict(out, batch)
        assert {"rgb_loss"} <= set(loss) and all(torch.isfinite(v).all() for v in loss.values()), loss
        if "mono" in name:
            assert {"depth_loss", "normal_loss"} <= set(loss)
        opts.zero_grad_all()
        sum(loss.values()).backward()
        opts.optimizer_scaler_step_all(torch.amp.GradScaler("cuda", enabled=False))  # engine/trainer.py:320-324
        opts.scheduler_step_all(step)
        for cb in callbacks:
            cb.run_callback_at_location(step, TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
    moved = [k for k, p in list(model.field.named_parameters())[:4] if not torch.equal(p.detach(), before[k])]
    assert moved, "two optimiser steps must move the field's parameters"


def test_background_fields_density_fn(device):
    """Field.density_fn (fields/base_field.py:48-65) of the two background fields: the density at explicit positions equals get_density on
    ray samples whose frustum mid points are those positions."""
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.fields.nerfacto_field import TCNNNerfactoField
    from sdfstudio_amd.fields.vanilla_nerf_field import NeRFEncoding, NeRFField
    from sdfstudio_amd.models.neus_facto import SceneContraction

    torch.manual_seed(0)
    n, s = 16, 6
    o, d = torch.randn(n, 3, device=device) * 0.3, torch.nn.functional.normalize(torch.randn(n, 3, device=device), dim=-1)
    starts = torch.sort(torch.rand(n, s, device=device) * 3.0, dim=-1)[0]
    rb = RayBundle(origins=o, directions=d, camera_indices=torch.zeros(n, 1, dtype=torch.long, device=device), nears=torch.zeros(n, 1, device=device),
                   fars=torch.full((n, 1), 4.0, device=device))
    rs = rb.get_ray_samples(starts, starts + 0.2)
    pos = rs.frustums.get_positions()
    sc = SceneContraction(order=float("inf"))
    fields = [NeRFField(position_encoding=NeRFEncoding(in_dim=3, num_frequencies=10, min_freq_exp=0.0, max_freq_exp=9.0, include_input=True),
                        direction_encoding=NeRFEncoding(in_dim=3, num_frequencies=4, min_freq_exp=0.0, max_freq_exp=3.0, include_input=True),
                        spatial_distortion=sc).to(device),
              TCNNNerfactoField(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=3, num_levels=4, max_res=32, log2_hashmap_size=8,
                                spatial_distortion=sc).to(device)]
    with torch.no_grad():
        for f in fields:
            dens, _ = f.get_density(rs)
            assert_close(type(f).__name__ + ".density_fn", f.density_fn(pos), dens, rtol=1e-5, atol=1e-7)
    with pytest.raises(NotImplementedError):
        NeRFField()  # the reference's Identity encodings are not built: refused, not replaced
