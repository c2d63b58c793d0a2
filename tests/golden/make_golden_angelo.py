"""Mint the golden vectors of BASELINE config 5's step (neus-facto-angelo) by running the reference's own model class.

    python tests/golden/make_golden_angelo.py      # writes tests/golden/neus_facto_angelo_small_train.npz   (build container only)

The reference's ``NeuSFactoModel`` (nerfstudio/models/neus_facto.py, unmodified, imported through oracle/ref_harness.py with the
documented PyTorch tinycudann shim) is set up exactly as the ``neus-facto-angelo`` preset does (configs/method_configs.py:381-450:
one-hidden-layer 256-wide geometry network on a 16-level x 8-feature LINEAR hash grid without positional encoding, numerical SDF
gradients, appearance embedding, ``background_model="grid"``, near 0.01 / far 1000 with the L-inf scene contraction, eikonal 0.01,
curvature loss 5e-4) with three things shrunk so that the file stays small: the hash tables (2^10 entries per level; the background
field's through its own constructor arguments, as make_golden_bg.py does), the sample counts (32 / 24 proposal -> 12 field samples)
and the batch (40 rays).  It runs in TRAIN mode - per-camera appearance embeddings in both fields, stratified draws injected through
``torch.rand`` - with the state the training callbacks (neus_facto.py:187-282) put the model in at some step: progressive level mask
(two files: ``level`` 8 of 16 = the preset's level_init, and all 16 = the steady state 85 % of the schedule runs in), the matching
numerical-gradient delta, a curvature-loss factor, cos / proposal anneal.  Recorded: inputs, the complete state_dict, ``model(ray_bundle)``
outputs + the field's per-sample outputs, the reference's OWN ``get_loss_dict`` (rgb L1, eikonal, interlevel, curvature) and the
gradient of the summed loss w.r.t. every parameter.  The script asserts that the oracle (oracle/sdf_path.py: neus_facto_forward with
``numerical_delta`` and ``background``, neus_facto_loss with ``curvature``) reproduces all of it before writing.
"""
import functools
import os
import sys

sys.dont_write_bytecode = True  # /root/reference is read-only: importing it must leave no __pycache__ there

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_harness, sdf_path as O  # noqa: E402
from helpers import (ANGELO_GOLDEN_BG, ANGELO_GOLDEN_LOG2_T, ANGELO_GOLDEN_N_FIELD, ANGELO_GOLDEN_PROPS, angelo_bg_levels,  # noqa: E402
                     angelo_oracle_cfg, assert_grads_close_mod_relu_flips, oracle_params_from_reference_state, relu_flip_basis)

torch.set_float32_matmul_precision("highest")

LOG2_T = ANGELO_GOLDEN_LOG2_T
N_RAYS = 40
COS_ANNEAL, ANNEAL, CURV_FACTOR = 0.35, 0.6, 0.7
PROPS, BG, N_FIELD = ANGELO_GOLDEN_PROPS, ANGELO_GOLDEN_BG, ANGELO_GOLDEN_N_FIELD
bg_levels = angelo_bg_levels
oracle_params = oracle_params_from_reference_state


def perturb(model, seed):
    """Noise on every parameter so that no path is dead (geometric init zeroes the first layer's grid-feature columns)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if not p.requires_grad or p.numel() == 1:
                continue
            if k == "field.encoding.params":
                # 1 / f spectrum: every level contributes a comparable d sdf / dx (scale_l x amplitude_l = const)
                lv = angelo_oracle_cfg().field.grid_levels()
                t = (torch.rand(p.shape, generator=g) * 2 - 1).view(-1, 8)
                for l in range(lv.n_levels):
                    t[int(lv.offset[l]):int(lv.offset[l + 1])] *= 0.3 * float(lv.scale[0]) / float(lv.scale[l])
                p.copy_(t.reshape(-1))
            elif k.endswith("encoding.params"):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.3)
            elif k.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "embedding" in k:
                p.mul_(0.3)
            else:
                p.add_(0.03 * torch.randn(p.shape, generator=g))


class _RandQueue:
    """Replays preset tensors for torch.rand so the reference's stratified jitter is reproducible."""

    def __init__(self, items):
        self.items = [t.clone() for t in items]
        self._orig = torch.rand

    def __enter__(self):
        def fake(*size, **kw):
            t = self.items.pop(0)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def main():
    ns = ref_harness.import_reference()
    import nerfstudio.fields.nerfacto_field as rnff
    import nerfstudio.models.base_surface_model as rbsm
    import nerfstudio.models.neus_facto as rnf
    from nerfstudio.data.scene_box import SceneBox

    H = ns.FieldHeadNames
    rbsm.TCNNNerfactoField = functools.partial(rnff.TCNNNerfactoField, **BG)
    sb = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5, radius=1.0, collider_type="near_far")
    fcfg = ns.sf.SDFFieldConfig(use_grid_feature=True, num_layers=1, num_layers_color=4, hidden_dim=256, hidden_dim_color=256,
                                geometric_init=True, bias=0.5, beta_init=0.3, inside_outside=False, use_appearance_embedding=True,
                                use_numerical_gradients=True, base_res=64, max_res=4096, log2_hashmap_size=LOG2_T,
                                hash_features_per_level=8, hash_smoothstep=False, use_position_encoding=False)
    mcfg = rnf.NeuSFactoModelConfig(near_plane=0.01, far_plane=1000.0, overwrite_near_far_plane=True, sdf_field=fcfg,
                                    background_model="grid", level_init=8, eikonal_loss_mult=0.01, use_anneal_beta=True,
                                    enable_progressive_hash_encoding=True, enable_numerical_gradients_schedule=True,
                                    enable_curvature_loss_schedule=True, curvature_loss_multi=5e-4, num_proposal_samples_per_ray=(32, 24),
                                    num_neus_samples_per_ray=N_FIELD, proposal_net_args_list=PROPS)
    cfg = angelo_oracle_cfg()
    n = N_RAYS
    o, d, cam = O.synthetic_rays(n, seed=23)
    o = o * 0.45  # cameras just outside the unit sphere (radius 1.23): with far = 1000 most of a DTU-distance ray's samples are background
    g = torch.Generator().manual_seed(5)
    image = torch.rand(n, 3, generator=g)
    rand = [torch.rand(n, 1, generator=g) for _ in range(3)]
    for level in (8, 16):
        torch.manual_seed(0)
        model = mcfg.setup(scene_box=sb, num_train_data=49, world_size=1, local_rank=0)
        perturb(model, seed=9)
        model.train()
        fld = model.field
        # the state neus_facto.py:187-282's callbacks leave the model in when `level` levels are active
        delta = 4.0 / (fld.base_res * fld.growth_factor ** (level - 1))  # :222-238 at step = (level - 1) steps_per_level, x 4 (:231-233)
        fld.update_mask(level)
        fld.set_numerical_gradients_delta(delta)
        fld.set_cos_anneal_ratio(COS_ANNEAL)
        model.proposal_sampler.set_anneal(ANNEAL)
        model.curvature_loss_multi_factor = CURV_FACTOR
        rb = ns.rays.RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
                               camera_indices=cam[:, None])
        with _RandQueue(rand) as rq:
            out = model(rb)
            assert not rq.items, "the sampler drew fewer tensors than injected"
        losses = model.get_loss_dict(out, {"image": image})
        assert set(losses) == {"rgb_loss", "eikonal_loss", "interlevel_loss", "curvature_loss"}, sorted(losses)
        model.zero_grad()
        sum(losses.values()).backward()
        fo, rs = out["field_outputs"], out["ray_samples"]
        rec = {
            "starts": rs.frustums.starts[..., 0], "ends": rs.frustums.ends[..., 0],
            "bins": torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], -1),
            "sdf": fo[H.SDF][..., 0], "gradient": fo[H.GRADIENT], "field_rgb": fo[H.RGB], "alpha": fo[H.ALPHA][..., 0],
            "sampled_sdf": fo["sampled_sdf"], "points_norm": fo["points_norm"][..., 0],
            "weights": out["weights"][..., 0], "rgb": out["rgb"], "depth": out["depth"][..., 0], "normal": out["normal"],
            "accumulation": out["accumulation"][..., 0],
            "prop_weights0": out["weights_list"][0][..., 0], "prop_weights1": out["weights_list"][1][..., 0],
        }
        inside = (rs.frustums.get_start_positions().norm(dim=-1) < 1.0)
        print(f"[level {level}] delta {delta:.5f}; {inside.float().mean().item():.2f} of the samples inside the unit sphere; losses "
              + ", ".join(f"{k} {v.item():.5f}" for k, v in losses.items()))
        assert 0.2 < inside.float().mean().item() < 0.9, "both branches of the fg / bg merge must be exercised"
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        assert all(grads[k].abs().max() > 0 for k in grads if k.startswith("field_background") or k == "field.encoding.params")
        tg = grads["field.encoding.params"].view(-1, 8)
        lv = cfg.field.grid_levels()
        if level < 16:
            assert tg[int(lv.offset[level]):].abs().max().item() == 0.0

        # ---- the oracle on the same inputs must reproduce the reference
        p = oracle_params(sd)
        mask = fld.hash_encoding_mask.clone()
        curv_mult = 5e-4 * CURV_FACTOR

        def oracle_step():
            po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
            oo = O.neus_facto_forward(o, d, cam, po, cfg, anneal=ANNEAL, cos_anneal_ratio=COS_ANNEAL, rand=rand, mask=mask, training=True,
                                      numerical_delta=delta, background={"prefix": "field_background.", "lv": bg_levels()})
            ol = O.neus_facto_loss(oo, image, cfg, curvature=(delta, curv_mult))
            sum(ol.values()).backward()
            return oo, ol, {k: v.grad for k, v in po.items() if v.grad is not None}

        oo, ol, _ = oracle_step()
        omap = {"starts": oo["starts"], "ends": oo["ends"], "bins": oo["bins"], "sdf": oo["field"]["sdf"], "gradient": oo["field"]["gradient"],
                "field_rgb": oo["field"]["rgb"], "alpha": oo["field"]["alpha"], "sampled_sdf": oo["field"]["sampled_sdf"],
                "points_norm": oo["field"]["points_norm"], "weights": oo["weights"], "rgb": oo["rgb"], "depth": oo["depth"],
                "normal": oo["normal"], "accumulation": oo["accumulation"], "prop_weights0": oo["weights_list"][0],
                "prop_weights1": oo["weights_list"][1]}
        worst = 0.0
        for k, v in rec.items():
            err = (omap[k].detach() - v.detach()).abs().max().item()
            scale = v.detach().abs().max().item() + 1e-12
            worst = max(worst, err / scale)
            # the finite-difference normal divides the sdf's fp32 round-off (~1e-6, summation order) by 2 delta; alpha, weights and
            # everything rendered see it through the cosine of sdf_field.py:494-516
            fd = 5e-7 / delta
            exact = ("starts", "ends", "bins", "sdf", "sampled_sdf", "points_norm", "prop_weights0", "prop_weights1")
            tol = max(1e-4, 4 * fd) if k in ("depth", "gradient", "normal") else (2e-5 if k in exact else max(2e-5, fd))
            assert err <= tol * scale + 1e-6, f"oracle != reference on {k}: abs {err:.3e} (scale {scale:.3e})"
        for k in losses:
            assert abs(ol[k].item() - losses[k].item()) <= 2e-5 * abs(losses[k].item()) + 1e-8, (k, ol[k].item(), losses[k].item())
        # gradients: equal up to the branch choices at the path's knife edges (colour-network ReLUs fed by the finite-difference
        # normal, sign of curvature elements below the second difference's round-off 4e-7 / delta^2)
        pg = oracle_params(grads)
        base, basis = relu_flip_basis(lambda: oracle_step()[2], margin=2e-5, curv_margin=1e-6 / (delta * delta), max_flips=64)
        assert_grads_close_mod_relu_flips({k: v for k, v in base.items() if k in pg}, pg, basis, rtol=2e-3)
        print(f"[level {level}] oracle reproduces the reference ({len(pg)} parameter gradients); worst forward rel err {worst:.2e}")

        blob = {"in/origins": o, "in/dirs": d, "in/cam": cam, "in/image": image, "in/level": torch.tensor(level),
                "in/delta": torch.tensor(delta, dtype=torch.float64), "in/cos_anneal": torch.tensor(COS_ANNEAL), "in/anneal": torch.tensor(ANNEAL),
                "in/curv_mult": torch.tensor(5e-4 * CURV_FACTOR, dtype=torch.float64)}
        for i, r in enumerate(rand):
            blob[f"in/rand{i}"] = r
        for k, v in sd.items():
            blob[f"param/{k}"] = v
        for k, v in rec.items():
            blob[f"out/{k}"] = v.detach()
        for k, v in losses.items():
            blob[f"loss/{k}"] = v.detach()
        for k, v in grads.items():
            blob[f"grad/{k}"] = v
        path = os.path.join(HERE, f"neus_facto_angelo_small_train_l{level}.npz")
        np.savez_compressed(path, **{k: v.numpy() for k, v in blob.items()})
        print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
