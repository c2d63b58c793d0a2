"""The ref-nerf colour options and the off-axis position encoding on the CPU: the oracle
against the golden vectors minted from the REFERENCE's own SDFField (tests/golden/make_golden_refnerf.py), live against the reference when
its tree is here, and the host-side behaviour of the product's field (construction, state_dict keys, refusals)."""
import os

import pytest
import torch

from helpers import load_golden_file
from oracle import sdf_path as O

CASES = ["diffuse", "tint", "reflections", "n_dot_v", "off_axis", "all"]
FLAGS = ("use_diffuse_color", "use_specular_tint", "use_reflections", "use_n_dot_v", "off_axis", "use_appearance_embedding")


def golden_cfg(g) -> O.FieldCfg:
    kw = {f: bool(v) for f, v in zip(FLAGS, g["misc"]["flags"].tolist())}
    return O.FieldCfg(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64, bias=0.5, inside_outside=False,
                      use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11,
                      hash_features_per_level=2, hash_smoothstep=True, skip_in=(), position_encoding_max_degree=int(g["misc"]["pe_degree"]), **kw)


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_the_references_refnerf_field(case):
    g = load_golden_file(f"sdf_field_refnerf_{case}.npz")
    cfg = golden_cfg(g)
    i = g["in"]
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in g["param"].items()}
    out = O.field_outputs(i["origins"], i["dirs"], i["starts"], i["ends"] - i["starts"], i["cam"], p, cfg)
    for k in ("rgb", "sdf", "gradient", "normal"):
        assert float((out[k] - g["out"][k]).abs().max()) < 2e-5, (case, k)
    loss = (out["rgb"] * i["c1"]).sum() + (out["sdf"] * i["c2"]).sum() + ((out["gradient"] ** 2) * i["c3"]).sum()
    assert float(loss) == pytest.approx(float(g["loss"]["total"]), rel=1e-4, abs=1e-4)
    loss.backward()
    heads = [k for k in g["grad"] if "_pred" in k]
    assert len(heads) == (2 if cfg.use_diffuse_color else 0) + (2 if cfg.use_specular_tint and cfg.use_diffuse_color else 0), heads
    for k, ref in g["grad"].items():
        if p[k].grad is None:
            assert float(ref.abs().max()) == 0.0, k
            continue
        assert float((p[k].grad - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-9, (case, k)


def test_product_field_constructs_with_the_options_and_keeps_the_references_state_dict_keys():
    """Host side only (no kernel runs on CPU tensors): parameter names and shapes are the reference's for every option - what a checkpoint
    trained by the reference needs."""
    from sdfstudio_amd.fields.sdf_field import SDFField, SDFFieldConfig

    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    f = SDFField(SDFFieldConfig(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64, use_grid_feature=True,
                                num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11, position_encoding_max_degree=1, off_axis=True,
                                use_diffuse_color=True, use_specular_tint=True, use_reflections=True, use_n_dot_v=True), aabb, 49)
    sd = f.state_dict()
    assert tuple(sd["diffuse_color_pred.weight"].shape) == (3, 64) and tuple(sd["specular_tint_pred.bias"].shape) == (3,)
    assert tuple(sd["glin0.weight_v"].shape) == (64, 3 + 42 + 16)           # off-axis encoding: 21 directions x 1 frequency x 2
    assert tuple(sd["clin0.weight_v"].shape) == (64, 27 + 64 + 32 + 1)      # use_diffuse_color: no position, no gradient; + n . v
    g = load_golden_file("sdf_field_refnerf_all.npz")
    assert {k for k in g["param"]} <= set(sd) | {"laplace_density.beta_min"}
    if os.path.isdir("/root/reference/nerfstudio"):
        from oracle import ref_harness

        ns = ref_harness.import_reference()
        ref = ns.sf.SDFField(ns.sf.SDFFieldConfig(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64,
                                                  use_grid_feature=True, num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11,
                                                  position_encoding_max_degree=1, off_axis=True, use_diffuse_color=True, use_specular_tint=True,
                                                  use_reflections=True, use_n_dot_v=True), aabb, 49)
        rsd = ref.state_dict()
        assert set(rsd) == set(sd), set(rsd) ^ set(sd)
        assert all(tuple(rsd[k].shape) == tuple(sd[k].shape) for k in rsd), [k for k in rsd if tuple(rsd[k].shape) != tuple(sd[k].shape)]


def test_periodic_encoding_without_grid_features_is_accepted():
    """encoding_type = "periodic", use_grid_feature = False - the configuration SURVEY 8(c) probed config 1 with.  The
    reference then builds a PeriodicVolumeEncoding it never evaluates (zero feature block, sdf_field.py:389-390); the product carries its
    `hash_table` for state_dict and refuses only the combination the reference itself cannot run."""
    from sdfstudio_amd.fields.sdf_field import SDFField, SDFFieldConfig

    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    f = SDFField(SDFFieldConfig(encoding_type="periodic", use_grid_feature=False), aabb, 49)
    sd = f.state_dict()
    assert tuple(sd["encoding.hash_table"].shape) == ((1 << 18) * 16, 2) and "encoding.params" not in sd
    assert f.encoding.n_output_dims == 32 and tuple(sd["glin0.weight_v"].shape) == (256, 3 + 36 + 32)
    if os.path.isdir("/root/reference/nerfstudio"):
        from oracle import ref_harness

        ns = ref_harness.import_reference()
        ref = ns.sf.SDFField(ns.sf.SDFFieldConfig(encoding_type="periodic", use_grid_feature=False), aabb, 49)
        rsd = ref.state_dict()
        assert set(rsd) == set(sd), set(rsd) ^ set(sd)
        f.load_state_dict(rsd)  # a reference checkpoint of that configuration loads
    with pytest.raises(NotImplementedError, match="periodic"):
        SDFField(SDFFieldConfig(encoding_type="periodic", use_grid_feature=True), aabb, 49)
    with pytest.raises(NotImplementedError, match="numerical"):
        SDFField(SDFFieldConfig(use_reflections=True, use_numerical_gradients=True), aabb, 49)
