"""The ref-nerf colour options and the off-axis position encoding ON THE GPU (-m gpu): sdfstudio_amd's SDFField with the flags against the
golden vectors of the REFERENCE's own SDFField (tests/golden/make_golden_refnerf.py) - forward heads and EVERY parameter gradient, the
four flags one at a time and all together, with the off-axis encoding; plus the periodic / no-grid-feature configuration against the oracle.
Bars: 1e-5 on sdf, 1e-4 of the tensor's scale on rgb / gradient (1e-3 element-wise), 2e-3 of the gradient's scale on parameter gradients."""
import pytest
import torch

from helpers import assert_close, load_golden_file
from oracle import sdf_path as O
from test_cpu_refnerf import CASES, golden_cfg
from test_gpu_parity import _bundle

pytestmark = pytest.mark.gpu


def _product_field(g, cfg, device, **extra):
    from sdfstudio_amd.fields.sdf_field import SDFField, SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import SceneContraction

    fcfg = SDFFieldConfig(num_layers=cfg.num_layers, hidden_dim=cfg.hidden_dim, geo_feat_dim=cfg.geo_feat_dim, num_layers_color=cfg.num_layers_color,
                          hidden_dim_color=cfg.hidden_dim_color, bias=cfg.bias, inside_outside=cfg.inside_outside, use_grid_feature=cfg.use_grid_feature,
                          beta_init=cfg.beta_init, num_levels=cfg.num_levels, max_res=cfg.max_res, base_res=cfg.base_res,
                          log2_hashmap_size=cfg.log2_hashmap_size, hash_features_per_level=cfg.hash_features_per_level,
                          hash_smoothstep=cfg.hash_smoothstep, position_encoding_max_degree=cfg.position_encoding_max_degree,
                          use_appearance_embedding=cfg.use_appearance_embedding, use_diffuse_color=cfg.use_diffuse_color,
                          use_specular_tint=cfg.use_specular_tint, use_reflections=cfg.use_reflections, use_n_dot_v=cfg.use_n_dot_v,
                          off_axis=cfg.off_axis, **extra)
    fld = SDFField(fcfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 49, spatial_distortion=SceneContraction(order=float("inf")))
    sd = fld.state_dict()
    for k, v in g["param"].items():
        if k in sd:
            assert tuple(sd[k].shape) == tuple(v.shape), (k, tuple(sd[k].shape), tuple(v.shape))
            sd[k] = v.clone()
    fld.load_state_dict(sd)
    return fld.to(device).train()


def _oracle_run(g, cfg, dtype=torch.float32):
    """() -> {name: gradient} of the test's loss through the oracle (pinned on the same golden by tests/test_cpu_refnerf.py)."""
    i = g["in"]
    cast = (lambda t: t.to(dtype) if t.is_floating_point() else t)

    def run():
        p = {k: cast(v).clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
        out = O.field_outputs(cast(i["origins"]), cast(i["dirs"]), cast(i["starts"]), cast(i["ends"] - i["starts"]), i["cam"], p, cfg)
        loss = (out["rgb"] * cast(i["c1"])).sum() + (out["sdf"] * cast(i["c2"])).sum() + ((out["gradient"] ** 2) * cast(i["c3"])).sum()
        loss.backward()
        run.out = {k: v.detach() for k, v in out.items() if torch.is_tensor(v)}
        return {k: v.grad for k, v in p.items() if v.grad is not None}

    return run


@pytest.mark.parametrize("case", CASES)
def test_refnerf_field_against_the_references_golden(case, device):
    from helpers import assert_fp32_class, assert_grads_close_mod_relu_flips, relu_flip_basis
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H

    g = load_golden_file(f"sdf_field_refnerf_{case}.npz")
    cfg = golden_cfg(g)
    i, ref = g["in"], g["out"]
    fld = _product_field(g, cfg, device)
    rs = _bundle(i["origins"], i["dirs"], i["cam"], 0.5, 4.5, device).get_ray_samples(i["starts"].to(device), i["ends"].to(device))
    out = fld(rs)
    # ---- forward heads against the REFERENCE's values.  sdf and rgb at the north-star bars; d sdf / dx sums ~60 first-layer columns of
    # both signs (42 of them the off-axis encoding's), so its bar is the fp32 class itself: as close to the fp64 evaluation as the
    # reference's own fp32 run is, times 3
    assert_close("sdf", out[H.SDF][..., 0], ref["sdf"], rtol=0, atol=1e-5)
    assert_close("rgb", out[H.RGB], ref["rgb"], rtol=1e-4, atol=2e-6)
    run64 = _oracle_run(g, cfg, torch.float64)
    run64()
    assert_fp32_class("gradient", out[H.GRADIENT], ref["gradient"], run64.out["gradient"], factor=3.0, atol=2e-5)
    assert_fp32_class("normal", out[H.NORMAL], ref["normal"], run64.out["normal"], factor=3.0, atol=2e-5)
    assert_close("gradient (loose, vs the reference)", out[H.GRADIENT], ref["gradient"], rtol=5e-4, atol=2e-6, elem_rtol=1e-2)
    loss = ((out[H.RGB] * i["c1"].to(device)).sum() + (out[H.SDF][..., 0] * i["c2"].to(device)).sum() +
            ((out[H.GRADIENT] ** 2) * i["c3"].to(device)).sum())
    assert float(loss.detach()) == pytest.approx(float(g["loss"]["total"]), rel=2e-4, abs=2e-4)
    loss.backward()
    got = {k: p.grad.detach().cpu() for k, p in fld.named_parameters() if p.grad is not None}
    # ---- every parameter gradient: against the oracle (which the CPU suite pins on this golden's gradients at 2e-3), modulo the ReLU
    # branch of colour-network pre-activations within 2e-6 of zero (helpers.relu_flip_basis: two correct fp32 evaluations may differ there)
    # With use_reflections / use_n_dot_v the colour inputs are functions of the NORMAL: the fp32 round-off of d sdf / dx (1e-4 of its scale:
    # the fp32-class bar above) enters the direction encoding multiplied by its highest frequency (8), so the colour network's
    # pre-activations carry ~1e-3 of input noise in ANY fp32 evaluation, the reference's included: the knife-edge margin follows it.
    sensitive = cfg.use_reflections or cfg.use_n_dot_v
    base, basis = relu_flip_basis(_oracle_run(g, cfg), margin=3e-4 if sensitive else 2e-6)
    keys = [k for k in base if float(base[k].abs().max()) > 0.0]
    assert len(keys) >= 15 and all(k in got for k in keys), [k for k in keys if k not in got]
    assert_grads_close_mod_relu_flips(got, {k: base[k] for k in keys}, basis, rtol=5e-3 if sensitive else 2e-3)
    # ... and, loosely, against the reference's own gradients in the golden: relative L2 error (a flipped ReLU moves single elements of a
    # gradient by more than any max-norm bar, in the reference's own fp32 as much as here)
    for k, r in g["grad"].items():
        if float(r.abs().max()) > 0.0:
            assert float((got[k] - r).norm()) <= 2e-2 * float(r.norm()) + 1e-7, k
    if cfg.use_diffuse_color:
        assert "diffuse_color_pred.weight" in got and "diffuse_color_pred.bias" in got
    if cfg.use_diffuse_color and cfg.use_specular_tint:
        assert "specular_tint_pred.weight" in got
    # eval mode (no graph): the same heads from the nothing-saved kernels
    fld.eval()
    with torch.no_grad():
        ev = fld(rs)
    if not cfg.use_appearance_embedding:  # (with the embedding the eval path feeds zeros, sdf_field.py:557-564: another input)
        assert_close("eval rgb", ev[H.RGB], ref["rgb"], rtol=1e-4, atol=2e-6)
    assert_close("eval sdf", ev[H.SDF][..., 0], ref["sdf"], rtol=0, atol=1e-5)


def test_refnerf_gradients_repeat_bit_for_bit(device):
    """The diffuse / tint heads' gradients come out of fixed-order reductions (csrc/refnerf_kernels.h: no atomics): two identical backward
    passes give the same bits."""
    g = load_golden_file("sdf_field_refnerf_all.npz")
    cfg = golden_cfg(g)
    i = g["in"]
    fld = _product_field(g, cfg, device)
    rs = _bundle(i["origins"], i["dirs"], i["cam"], 0.5, 4.5, device).get_ray_samples(i["starts"].to(device), i["ends"].to(device))
    runs = []
    for _ in range(3):
        fld.zero_grad()
        sdf, grad, rgb, _ = fld.forward_fused(rs)
        ((rgb * i["c1"].to(device)).sum() + (grad ** 2).sum()).backward()
        runs.append({k: p.grad.detach().clone() for k, p in fld.named_parameters() if p.grad is not None and "encoding" not in k and "embedding" not in k})
    for r in runs[1:]:
        for k, v in runs[0].items():
            assert torch.equal(r[k], v), k


def test_periodic_encoding_without_grid_features_runs_like_the_pure_mlp(device):
    """encoding_type = "periodic", use_grid_feature = False computes what the hash configuration without grid features
    computes - the feature block is zeros either way (sdf_field.py:389-390) - and matches the oracle."""
    from sdfstudio_amd.fields.field_heads import FieldHeadNames as H
    from sdfstudio_amd.fields.sdf_field import SDFField, SDFFieldConfig

    torch.manual_seed(3)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    kw = dict(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64, num_levels=8, max_res=128, base_res=4,
              log2_hashmap_size=11, bias=0.5, inside_outside=False, use_grid_feature=False)
    a = SDFField(SDFFieldConfig(encoding_type="periodic", **kw), aabb, 49).to(device).train()
    b = SDFField(SDFFieldConfig(encoding_type="hash", **kw), aabb, 49).to(device).train()
    sd = {k: v for k, v in a.state_dict().items() if not k.startswith("encoding.")}
    b.load_state_dict({**b.state_dict(), **sd})
    o, d, cam = O.synthetic_rays(32, seed=2)
    starts = torch.sort(torch.rand(32, 5) * 3.0 + 0.6, dim=-1)[0]
    rs = _bundle(o, d, cam, 0.5, 4.5, device).get_ray_samples(starts.to(device), starts.to(device) + 0.05)
    oa, ob = a(rs), b(rs)
    for k in (H.SDF, H.RGB, H.GRADIENT):
        assert torch.equal(oa[k], ob[k]), k
    (oa[H.RGB].sum() + oa[H.SDF].sum()).backward()
    assert a.encoding.hash_table.grad is None  # never evaluated, as in the reference
    assert all(torch.isfinite(p.grad).all() for n, p in a.named_parameters() if p.grad is not None)
    cfg = O.FieldCfg(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64, bias=0.5, inside_outside=False,
                     use_grid_feature=False, num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11, skip_in=())
    p = {k: v.detach().cpu() for k, v in a.state_dict().items()}
    p["encoding.params"] = torch.zeros(1)
    ref = O.field_outputs(o, d, starts, torch.full_like(starts, 0.05), cam, p, cfg)
    # (the product field above has no spatial distortion, the oracle contracts: positions inside the unit box are fixed points of it)
    inside = (o[:, None, :] + d[:, None, :] * starts[..., None]).abs().amax(-1) < 1.0
    assert_close("sdf vs oracle", oa[H.SDF][..., 0].detach().cpu()[inside], ref["sdf"][inside], rtol=0, atol=1e-5)
