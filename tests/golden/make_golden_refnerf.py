"""Golden vectors for the ref-nerf colour options and the off-axis position encoding of SDFField (fields/sdf_field.py:532-612,
field_components/encodings.py:118-208; the bakedsdf / bakedangelo field settings, configs/method_configs.py:270-286) from the REFERENCE's
own Python (build container only):

    python tests/golden/make_golden_refnerf.py     # writes tests/golden/sdf_field_refnerf_<case>.npz

Per case: the reference SDFField (imported unmodified through oracle/ref_harness.py, tcnn shim for the hash grid) is built with the flags at
the small test size, loaded with seeded perturbed parameters, evaluated on fixed ray samples in training mode (get_outputs: sdf, gradient,
normal, rgb) and differentiated: d (sum rgb c1 + sum sdf c2 + sum |grad|^2 c3) / d every parameter.  The oracle is asserted against the
reference on the spot (outputs and every gradient), then inputs / parameters / outputs / gradients are stored.

Cases: each flag alone, all four together (with appearance embedding and off_axis), and off_axis alone.  Position-encoding degree 1 with
off_axis: in0 = 3 + 42 + 16 = 61 columns, the width the small kernel family holds (two 32-column blocks); the bakedsdf preset's own degree 8
(371 columns) has no kernel instantiation (sdfstudio_amd/fields/sdf_field.py says so when asked).
"""
import os
import sys

sys.dont_write_bytecode = True  # /root/reference is read-only: importing it must leave no __pycache__ there

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness, sdf_path as O  # noqa: E402

torch.set_float32_matmul_precision("highest")

CASES = {
    "diffuse": dict(use_diffuse_color=True),
    "tint": dict(use_diffuse_color=True, use_specular_tint=True),
    "reflections": dict(use_reflections=True),
    "n_dot_v": dict(use_n_dot_v=True),
    "off_axis": dict(off_axis=True, position_encoding_max_degree=1),
    "all": dict(use_diffuse_color=True, use_specular_tint=True, use_reflections=True, use_n_dot_v=True, off_axis=True,
                position_encoding_max_degree=1, use_appearance_embedding=True),
}


def field_cfg(**kw) -> O.FieldCfg:
    base = dict(num_layers=2, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64, bias=0.5, inside_outside=False,
                use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11, hash_features_per_level=2,
                hash_smoothstep=True, skip_in=())
    base.update(kw)
    return O.FieldCfg(**base)


def perturbed(cfg: O.FieldCfg, seed: int):
    g = torch.Generator().manual_seed(seed + 100)
    p = O.init_field_params(cfg, num_images=49, seed=seed)
    for k in list(p):
        if k.endswith("weight_v") or k.endswith("_pred.weight"):
            p[k] = p[k] + 0.05 * torch.randn(p[k].shape, generator=g)
        elif k.endswith("weight_g"):
            p[k] = p[k] * (1.0 + 0.1 * torch.randn(p[k].shape, generator=g))
        elif k.endswith(".bias"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=g)
        elif k.endswith("encoding.params"):
            p[k] = (torch.rand(p[k].shape, generator=g) * 2 - 1) * 0.3
    return p


def main():
    ns = ref_harness.import_reference()
    H = ns.FieldHeadNames
    n, s = 48, 6
    for ci, (name, kw) in enumerate(CASES.items()):
        cfg = field_cfg(**kw)
        p = perturbed(cfg, seed=7 + ci)
        rcfg = ns.sf.SDFFieldConfig(
            num_layers=cfg.num_layers, hidden_dim=cfg.hidden_dim, geo_feat_dim=cfg.geo_feat_dim, num_layers_color=cfg.num_layers_color,
            hidden_dim_color=cfg.hidden_dim_color, bias=cfg.bias, inside_outside=cfg.inside_outside, use_grid_feature=True,
            beta_init=cfg.beta_init, num_levels=cfg.num_levels, max_res=cfg.max_res, base_res=cfg.base_res,
            log2_hashmap_size=cfg.log2_hashmap_size, hash_features_per_level=cfg.hash_features_per_level, hash_smoothstep=cfg.hash_smoothstep,
            position_encoding_max_degree=cfg.position_encoding_max_degree, use_appearance_embedding=cfg.use_appearance_embedding,
            use_diffuse_color=cfg.use_diffuse_color, use_specular_tint=cfg.use_specular_tint, use_reflections=cfg.use_reflections,
            use_n_dot_v=cfg.use_n_dot_v, off_axis=cfg.off_axis)
        field = ns.sf.SDFField(rcfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=49,
                               spatial_distortion=ns.sd.SceneContraction(order=float("inf")))
        field.skip_in = []  # 2 hidden layers: no skip (the reference hard-codes skip_in = [4] and never reaches it)
        sd = field.state_dict()
        missing = [k for k in sd if k not in p and k not in ("aabb",)]
        assert not missing, missing
        for k in sd:
            if k in p:
                assert sd[k].shape == p[k].shape, (k, sd[k].shape, p[k].shape)
                sd[k] = p[k].clone()
        field.load_state_dict(sd)
        field.train()
        o, d, cam = O.synthetic_rays(n, seed=11 + ci)
        gen = torch.Generator().manual_seed(3 + ci)
        starts = torch.sort(torch.rand(n, s, generator=gen) * 3.0 + 0.6, dim=-1)[0]
        ends = starts + 0.05
        rb = ns.rays.RayBundle(origins=o, directions=d, pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
                               camera_indices=cam[:, None], nears=torch.full((n, 1), 0.5), fars=torch.full((n, 1), 4.5))
        rs = rb.get_ray_samples(bin_starts=starts[..., None], bin_ends=ends[..., None])
        fo = field(rs)
        c1, c2, c3 = torch.randn(n, s, 3, generator=gen), torch.randn(n, s, generator=gen), torch.randn(n, s, 3, generator=gen) * 0.3
        loss = (fo[H.RGB] * c1).sum() + (fo[H.SDF][..., 0] * c2).sum() + ((fo[H.GRADIENT] ** 2) * c3).sum()
        loss.backward()
        ref_grads = {k: v.grad.detach().clone() for k, v in field.named_parameters() if v.grad is not None}
        # ---- the oracle on the same inputs
        po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
        oo = O.field_outputs(o, d, starts, ends - starts, cam, po, cfg)
        lo = (oo["rgb"] * c1).sum() + (oo["sdf"] * c2).sum() + ((oo["gradient"] ** 2) * c3).sum()
        lo.backward()
        for key, a, b in (("rgb", oo["rgb"], fo[H.RGB]), ("sdf", oo["sdf"], fo[H.SDF][..., 0]), ("gradient", oo["gradient"], fo[H.GRADIENT]),
                          ("normal", oo["normal"], fo[H.NORMAL])):
            err = float((a.detach() - b.detach()).abs().max())
            assert err < 2e-5, (name, key, err)
        worst = 0.0
        for k, gr in ref_grads.items():
            if po[k].grad is None:
                assert float(gr.abs().max()) == 0.0, (name, k)
                continue
            e = float((po[k].grad - gr).abs().max()) / (float(gr.abs().max()) + 1e-12)
            worst = max(worst, e)
            assert e < 2e-3, (name, k, e)
        save = {"in/origins": o, "in/dirs": d, "in/cam": cam, "in/starts": starts, "in/ends": ends, "in/c1": c1, "in/c2": c2, "in/c3": c3}
        for k, v in p.items():
            save["param/" + k] = v
        for key, v in (("rgb", fo[H.RGB]), ("sdf", fo[H.SDF][..., 0]), ("gradient", fo[H.GRADIENT]), ("normal", fo[H.NORMAL])):
            save["out/" + key] = v.detach()
        save["loss/total"] = loss.detach()
        for k, v in ref_grads.items():
            save["grad/" + k] = v
        flags = {f: bool(getattr(cfg, f)) for f in ("use_diffuse_color", "use_specular_tint", "use_reflections", "use_n_dot_v", "off_axis",
                                                     "use_appearance_embedding")}
        save["misc/flags"] = np.array([int(v) for v in flags.values()], np.int32)
        save["misc/pe_degree"] = np.int32(cfg.position_encoding_max_degree)
        np.savez_compressed(os.path.join(HERE, f"sdf_field_refnerf_{name}.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in save.items()})
        print(f"{name}: in0 {cfg.geo_in_dim()} colour in {cfg.color_in_dim()} loss {float(loss):.6f}  oracle worst rel grad err {worst:.2e}  "
              f"grads {len(ref_grads)}")


if __name__ == "__main__":
    main()
