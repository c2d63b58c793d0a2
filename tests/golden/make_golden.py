"""Mint the golden vectors by running the REFERENCE's own Python (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/neus_facto_small_{train,eval}.npz

What it does
  1. imports ``/root/reference/nerfstudio`` unmodified through ``oracle.ref_harness`` (stub modules for the
     absent third-party packages + the documented PyTorch ``tinycudann`` shim for the hash grid);
  2. builds the reference ``SDFField`` (fields/sdf_field.py), two ``HashMLPDensityField`` (fields/density_fields.py),
     ``ProposalNetworkSampler`` (model_components/ray_samplers.py:497), the renderers (model_components/renderers.py)
     and ``interlevel_loss_zip`` (model_components/losses.py:131), loads seeded parameters, and runs
     sample -> field -> weights -> render -> loss -> backward exactly as ``NeuSFactoModel`` does
     (models/neus_facto.py:282-310, models/base_surface_model.py:292-406);
  3. runs ``oracle.sdf_path`` on the same inputs and asserts it reproduces the reference (this pins the oracle);
  4. stores inputs, parameters, every output and every parameter gradient as a small ``.npz``.

The stratified-sampling draws (``torch.rand`` at ray_samplers.py:107,326) are injected so both sides see the same u.
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


def small_cfg() -> O.ModelCfg:
    f = O.FieldCfg(
        num_layers=8, hidden_dim=64, geo_feat_dim=64, num_layers_color=4, hidden_dim_color=64,
        bias=0.5, inside_outside=False, use_grid_feature=True, beta_init=0.3,
        num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11, hash_features_per_level=2, hash_smoothstep=True,
    )
    props = (
        O.ProposalCfg(hidden_dim=16, num_levels=5, max_res=32, base_res=4, log2_hashmap_size=9),
        O.ProposalCfg(hidden_dim=16, num_levels=5, max_res=64, base_res=4, log2_hashmap_size=9),
    )
    return O.ModelCfg(field=f, proposals=props, num_proposal_samples=(32, 24), num_neus_samples=16)


def perturbed_params(cfg: O.ModelCfg, seed=0):
    """Geometric init + noise everywhere, so that no input column / table entry is dead in the test."""
    g = torch.Generator().manual_seed(seed + 100)
    p = O.init_field_params(cfg.field, num_images=49, seed=seed)
    p.update(O.init_proposal_params(cfg.proposals, seed=seed + 1))
    for k in list(p.keys()):
        if k.endswith("weight_v"):
            p[k] = p[k] + 0.05 * torch.randn(p[k].shape, generator=g)
        elif k.endswith("weight_g"):
            p[k] = p[k] * (1.0 + 0.1 * torch.randn(p[k].shape, generator=g))
        elif k.endswith(".bias"):
            p[k] = p[k] + 0.02 * torch.randn(p[k].shape, generator=g)
        elif k.endswith("encoding.params") or k.endswith(".table"):
            p[k] = (torch.rand(p[k].shape, generator=g) * 2 - 1) * 0.3
    return p


def build_reference(ns, cfg: O.ModelCfg, p):
    fc = cfg.field
    rcfg = ns.sf.SDFFieldConfig(
        num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
        num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
        inside_outside=fc.inside_outside, use_grid_feature=True, beta_init=fc.beta_init, num_levels=fc.num_levels,
        max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
        hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep,
    )
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    contraction = ns.sd.SceneContraction(order=float("inf"))
    field = ns.sf.SDFField(rcfg, aabb, num_images=49, spatial_distortion=contraction)
    sd = field.state_dict()
    for k in sd:
        if k in p:
            assert sd[k].shape == p[k].shape, (k, sd[k].shape, p[k].shape)
            sd[k] = p[k].clone()
    field.load_state_dict(sd)
    nets = []
    for i, pc in enumerate(cfg.proposals):
        net = ns.df.HashMLPDensityField(
            aabb, spatial_distortion=contraction, hidden_dim=pc.hidden_dim, num_levels=pc.num_levels,
            max_res=pc.max_res, base_res=pc.base_res, log2_hashmap_size=pc.log2_hashmap_size,
            features_per_level=pc.features_per_level,
        )
        with torch.no_grad():
            net.mlp_base.encoding.params.copy_(p[f"proposal_networks.{i}.table"])
            net.mlp_base.w1.copy_(p[f"proposal_networks.{i}.w1"])
            net.mlp_base.w2.copy_(p[f"proposal_networks.{i}.w2"])
        nets.append(net)
    sampler = None
    if len(cfg.proposals) > 0:
        sampler = ns.rs.ProposalNetworkSampler(
            num_nerf_samples_per_ray=cfg.num_neus_samples, num_proposal_samples_per_ray=cfg.num_proposal_samples,
            num_proposal_network_iterations=len(cfg.proposals), single_jitter=True, update_sched=lambda step: -1,
        )
    return field, nets, sampler


class _RandQueue:
    """Replays preset tensors for torch.rand so the reference's stratified jitter is reproducible."""

    def __init__(self, items):
        self.items = list(items)
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


def run_reference(ns, field, nets, sampler, cfg, origins, dirs, cam, image, rand, training, cos_anneal, anneal):
    H = ns.FieldHeadNames
    for m in [field, sampler, *nets]:
        m.train(training)
    field.set_cos_anneal_ratio(cos_anneal)
    sampler.set_anneal(anneal)
    n = origins.shape[0]
    rb = ns.rays.RayBundle(
        origins=origins, directions=dirs, pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
        camera_indices=cam[:, None], nears=torch.full((n, 1), cfg.near), fars=torch.full((n, 1), cfg.far),
    )
    with _RandQueue(rand if training else []):
        ray_samples, weights_list, rs_list = sampler(rb, density_fns=[m.density_fn for m in nets])
    fo = field(ray_samples, return_alphas=True)
    weights = ray_samples.get_weights_from_alphas(fo[H.ALPHA])
    rgb_r = ns.rd.RGBRenderer(background_color=torch.zeros(3))
    rgb_r.train(training)
    rgb = rgb_r(rgb=fo[H.RGB], weights=weights)
    depth = ns.rd.DepthRenderer(method="expected")(weights=weights, ray_samples=ray_samples)
    normal = ns.rd.SemanticRenderer()(semantics=fo[H.NORMAL], weights=weights)
    acc = ns.rd.AccumulationRenderer()(weights=weights)
    out = {
        "starts": ray_samples.frustums.starts[..., 0], "ends": ray_samples.frustums.ends[..., 0],
        "bins": torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], -1),
        "sdf": fo[H.SDF][..., 0], "gradient": fo[H.GRADIENT], "field_rgb": fo[H.RGB], "alpha": fo[H.ALPHA][..., 0],
        "density": fo[H.DENSITY][..., 0], "field_normal": fo[H.NORMAL], "points_norm": fo["points_norm"][..., 0],
        "weights": weights[..., 0], "rgb": rgb, "depth": depth[..., 0], "normal": normal, "accumulation": acc[..., 0],
        "prop_weights0": weights_list[0][..., 0], "prop_weights1": weights_list[1][..., 0],
    }
    losses = {}
    if training:
        losses["rgb_loss"] = torch.nn.L1Loss()(image, rgb)
        losses["eikonal_loss"] = ((fo[H.GRADIENT].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
        losses["interlevel_loss"] = cfg.interlevel_loss_mult * ns.losses.interlevel_loss_zip(
            weights_list + [weights], rs_list + [ray_samples]
        )
    return out, losses


def main():
    ns = ref_harness.import_reference()
    cfg = small_cfg()
    p = perturbed_params(cfg)
    n = 64
    origins, dirs, cam = O.synthetic_rays(n, seed=42)
    g = torch.Generator().manual_seed(7)
    image = torch.rand(n, 3, generator=g)
    rand = [torch.rand(n, 1, generator=g) for _ in range(3)]
    cos_anneal, anneal = 0.3, 0.7
    field, nets, sampler = build_reference(ns, cfg, p)

    for mode in ("train", "eval"):
        training = mode == "train"
        out, losses = run_reference(ns, field, nets, sampler, cfg, origins, dirs, cam, image, rand, training,
                                    cos_anneal, anneal)
        ref_grads = {}
        if training:
            for m in [field, *nets]:
                m.zero_grad()
            total = sum(losses.values())
            total.backward()
            for k, v in field.named_parameters():
                if v.grad is not None:
                    ref_grads[k] = v.grad.clone()
            for i, m in enumerate(nets):
                ref_grads[f"proposal_networks.{i}.table"] = m.mlp_base.encoding.params.grad.clone()
                ref_grads[f"proposal_networks.{i}.w1"] = m.mlp_base.w1.grad.clone()
                ref_grads[f"proposal_networks.{i}.w2"] = m.mlp_base.w2.grad.clone()

        # ---- oracle on the same inputs; must reproduce the reference -------------------------------------
        po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min")
              for k, v in p.items()}
        o = O.neus_facto_forward(origins, dirs, cam, po, cfg, anneal=anneal, cos_anneal_ratio=cos_anneal,
                                 rand=rand if training else None, training=training)
        if not training:
            o["rgb"] = o["rgb"].clamp(0.0, 1.0)  # renderers.py:116-117
        omap = {
            "starts": o["starts"], "ends": o["ends"], "bins": o["bins"], "sdf": o["field"]["sdf"],
            "gradient": o["field"]["gradient"], "field_rgb": o["field"]["rgb"], "alpha": o["field"]["alpha"],
            "density": o["field"]["density"], "field_normal": o["field"]["normal"],
            "points_norm": o["field"]["points_norm"], "weights": o["weights"], "rgb": o["rgb"], "depth": o["depth"],
            "normal": o["normal"], "accumulation": o["accumulation"], "prop_weights0": o["weights_list"][0],
            "prop_weights1": o["weights_list"][1],
        }
        worst = 0.0
        for k, v in out.items():
            err = (omap[k].detach() - v.detach()).abs().max().item()
            scale = v.detach().abs().max().item() + 1e-12
            worst = max(worst, err / scale)
            # expected depth divides by the accumulated weight (renderers.py:255): ill-conditioned on empty rays
            tol = 1e-4 if k == "depth" else 2e-5
            assert err <= tol * scale + 1e-6, f"oracle != reference on {k}: abs {err:.3e} (scale {scale:.3e})"
        if training:
            ol = O.neus_facto_loss(o, image, cfg)
            for k in losses:
                assert abs(ol[k].item() - losses[k].item()) <= 1e-5 * abs(losses[k].item()) + 1e-8, k
            sum(ol.values()).backward()
            for k, gref in ref_grads.items():
                gor = po[k].grad
                err = (gor - gref).abs().max().item()
                scale = gref.abs().max().item() + 1e-12
                worst = max(worst, err / scale)
                assert err <= 1e-3 * scale + 1e-9, f"oracle grad != reference on {k}: {err:.3e} / {scale:.3e}"
        print(f"[{mode}] oracle reproduces the reference; worst rel err {worst:.2e}")

        blob = {"in/origins": origins, "in/dirs": dirs, "in/cam": cam, "in/image": image,
                "in/cos_anneal": torch.tensor(cos_anneal), "in/anneal": torch.tensor(anneal)}
        for i, r in enumerate(rand):
            blob[f"in/rand{i}"] = r
        for k, v in p.items():
            blob[f"param/{k}"] = v
        for k, v in out.items():
            blob[f"out/{k}"] = v.detach()
        for k, v in losses.items():
            blob[f"loss/{k}"] = v.detach()
        for k, v in ref_grads.items():
            blob[f"grad/{k}"] = v
        path = os.path.join(HERE, f"neus_facto_small_{mode}.npz")
        np.savez_compressed(path, **{k: v.numpy() for k, v in blob.items()})
        print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def main_neus():
    """NeuS (models/neus.py): NeuSSampler (ray_samplers.py:815) driving SDFField.get_sdf, then the same field / renderer path.
    Small sampler (16 + 4 x 4 samples) so the vectors stay small; writes tests/golden/neus_small_{train,eval}.npz."""
    ns = ref_harness.import_reference()
    cfg = small_cfg()
    p = {k: v for k, v in perturbed_params(cfg).items() if not k.startswith("proposal_networks.")}
    n = 48
    origins, dirs, cam = O.synthetic_rays(n, seed=43)
    g = torch.Generator().manual_seed(9)
    image = torch.rand(n, 3, generator=g)
    num_samples, num_importance, steps, base_var = 16, 16, 4, 64.0
    rand = [torch.rand(n, 1, generator=g) for _ in range(1 + steps)]
    cos_anneal = 0.3
    field, _, _ = build_reference(ns, O.ModelCfg(field=cfg.field, proposals=()), p)
    sampler = ns.rs.NeuSSampler(num_samples=num_samples, num_samples_importance=num_importance, num_samples_outside=0,
                                num_upsample_steps=steps, base_variance=base_var)
    H = ns.FieldHeadNames
    for mode in ("train", "eval"):
        training = mode == "train"
        for m in [field, sampler]:
            m.train(training)
        field.set_cos_anneal_ratio(cos_anneal)
        rb = ns.rays.RayBundle(
            origins=origins, directions=dirs, pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
            camera_indices=cam[:, None], nears=torch.full((n, 1), cfg.near), fars=torch.full((n, 1), cfg.far),
        )
        with _RandQueue(rand if training else []):
            ray_samples = sampler(rb, sdf_fn=field.get_sdf)
        # the same loop once more, step by step with the reference's own methods (ray_samplers.py:851-886), to record every
        # upsampling step's inputs and outputs: single steps are well conditioned and are compared tightly
        steps_blob = {}
        with _RandQueue(rand if training else []), torch.no_grad():
            sb = lambda r: torch.cat([r.spacing_starts[..., 0], r.spacing_ends[..., -1:, 0]], -1)
            rs_k = sampler.uniform_sampler(rb, num_samples=num_samples)
            new_k, sdf_k, idx_k = rs_k, None, None
            for it in range(steps):
                new_sdf = field.get_sdf(new_k)
                sdf_k = new_sdf if idx_k is None else torch.gather(torch.cat([sdf_k.squeeze(-1), new_sdf.squeeze(-1)], -1), 1,
                                                                     idx_k).unsqueeze(-1)
                al = sampler.rendering_sdf_with_fixed_inv_s(rs_k, sdf_k.reshape(rs_k.shape), inv_s=base_var * 2**it)
                w = rs_k.get_weights_from_alphas(al[..., None])
                w = torch.cat((w, torch.zeros_like(w[:, :1])), dim=1)
                new_k = sampler.pdf_sampler(rb, rs_k, w, num_samples=num_importance // steps)
                steps_blob[f"step{it}/bins_in"] = sb(rs_k)
                steps_blob[f"step{it}/sdf_in"] = sdf_k.reshape(rs_k.shape)
                steps_blob[f"step{it}/alpha"] = al
                steps_blob[f"step{it}/new_bins"] = sb(new_k)
                rs_k, idx_k = sampler.error_bounded_sampler.merge_ray_samples(rb, rs_k, new_k)
                steps_blob[f"step{it}/merged_bins"] = sb(rs_k)
                steps_blob[f"step{it}/index"] = idx_k
            assert torch.equal(sb(rs_k), sb(ray_samples)), "step-by-step replay differs from NeuSSampler.generate_ray_samples"
        fo = field(ray_samples, return_alphas=True)
        weights, trans = ray_samples.get_weights_and_transmittance_from_alphas(fo[H.ALPHA])
        rgb_r = ns.rd.RGBRenderer(background_color=torch.zeros(3))
        rgb_r.train(training)
        rgb = rgb_r(rgb=fo[H.RGB], weights=weights)
        depth = ns.rd.DepthRenderer(method="expected")(weights=weights, ray_samples=ray_samples)
        normal = ns.rd.SemanticRenderer()(semantics=fo[H.NORMAL], weights=weights)
        acc = ns.rd.AccumulationRenderer()(weights=weights)
        out = {
            "starts": ray_samples.frustums.starts[..., 0], "ends": ray_samples.frustums.ends[..., 0],
            "bins": torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], -1),
            "sdf": fo[H.SDF][..., 0], "gradient": fo[H.GRADIENT], "field_rgb": fo[H.RGB], "alpha": fo[H.ALPHA][..., 0],
            "weights": weights[..., 0], "rgb": rgb, "depth": depth[..., 0], "normal": normal, "accumulation": acc[..., 0],
        }
        # single upsampling steps of the oracle on the reference's inputs
        nears_t, fars_t = torch.full((n,), cfg.near), torch.full((n,), cfg.far)
        for it in range(steps):
            b_in, s_in = steps_blob[f"step{it}/bins_in"], steps_blob[f"step{it}/sdf_in"]
            eu = O.uniform_to_euclidean(b_in, nears_t, fars_t)
            al = O.neus_upsample_alpha(s_in, eu[:, 1:] - eu[:, :-1], base_var * 2**it)
            assert (al - steps_blob[f"step{it}/alpha"]).abs().max().item() <= 2e-6, f"step {it} alpha"
            wo, _ = O.weights_from_alphas(al)
            wo = torch.cat([wo, torch.zeros_like(wo[:, :1])], 1)
            nb = O.pdf_sample(wo, b_in, num_importance // steps, rand[1 + it] if training else None, histogram_padding=1e-5)
            mb, ix = O.merge_bins(b_in, nb)
            e1 = (nb - steps_blob[f"step{it}/new_bins"]).abs().max().item()
            assert e1 <= 2e-5, f"step {it} new bins {e1:.2e}"
            same = torch.equal(ix, steps_blob[f"step{it}/index"])
            print(f"[neus {mode}] step {it}: |d new bins| {e1:.1e}, merge index identical: {same}")
        losses, ref_grads = {}, {}
        if training:
            losses["rgb_loss"] = torch.nn.L1Loss()(image, rgb)
            losses["eikonal_loss"] = ((fo[H.GRADIENT].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
            field.zero_grad()
            sum(losses.values()).backward()
            for k, v in field.named_parameters():
                if v.grad is not None:
                    ref_grads[k] = v.grad.clone()
        # ---- oracle on the same inputs
        po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
        # (1) the sampler restatement: four rounds of inverse-CDF resampling with histogram_padding = 1e-5 divide by cdf
        # increments of ~1e-6, so fp32 round-off (1e-6 on the sdf) moves individual samples by up to a few 1e-3 after the
        # fourth round; the restatement must agree on the bulk of the samples and stay sorted / in range
        with torch.no_grad():
            o_s = O.neus_forward(origins, dirs, cam, p, cfg, cos_anneal_ratio=cos_anneal, rand=rand if training else None,
                                 training=training, num_samples=num_samples, num_samples_importance=num_importance,
                                 num_upsample_steps=steps, base_variance=base_var)
        d_bins = (o_s["bins"] - out["bins"]).abs()
        assert d_bins.median().item() <= 2e-6 and d_bins.max().item() <= 5e-3, (d_bins.median().item(), d_bins.max().item())
        print(f"[neus {mode}] sampler: median |d bins| {d_bins.median().item():.1e}, max {d_bins.max().item():.1e}")
        # (2) field + renderer + losses + gradients on IDENTICAL samples (the reference's)
        o = O.neus_forward(origins, dirs, cam, po, cfg, cos_anneal_ratio=cos_anneal, training=training,
                           samples=(out["bins"], out["starts"], out["ends"]))
        if not training:
            o["rgb"] = o["rgb"].clamp(0.0, 1.0)
        omap = {"starts": o["starts"], "ends": o["ends"], "bins": o["bins"], "sdf": o["field"]["sdf"],
                "gradient": o["field"]["gradient"], "field_rgb": o["field"]["rgb"], "alpha": o["field"]["alpha"],
                "weights": o["weights"], "rgb": o["rgb"], "depth": o["depth"], "normal": o["normal"],
                "accumulation": o["accumulation"]}
        worst = 0.0
        for k, v in out.items():
            err = (omap[k].detach() - v.detach()).abs().max().item()
            scale = v.detach().abs().max().item() + 1e-12
            worst = max(worst, err / scale)
            tol = 1e-4 if k == "depth" else 2e-5
            assert err <= tol * scale + 1e-6, f"[neus] oracle != reference on {k}: abs {err:.3e} (scale {scale:.3e})"
        if training:
            g_o = o["field"]["gradient"]
            ol = {"rgb_loss": torch.nn.functional.l1_loss(o["rgb"], image),
                  "eikonal_loss": ((g_o.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult}
            for k in losses:
                assert abs(ol[k].item() - losses[k].item()) <= 1e-5 * abs(losses[k].item()) + 1e-8, k
            sum(ol.values()).backward()
            for k, gref in ref_grads.items():
                err = (po[k].grad - gref).abs().max().item()
                scale = gref.abs().max().item() + 1e-12
                worst = max(worst, err / scale)
                assert err <= 1e-3 * scale + 1e-9, f"[neus] oracle grad != reference on {k}: {err:.3e} / {scale:.3e}"
        print(f"[neus {mode}] oracle reproduces the reference; worst rel err {worst:.2e}")
        blob = {"in/origins": origins, "in/dirs": dirs, "in/cam": cam, "in/image": image, "in/cos_anneal": torch.tensor(cos_anneal),
                "in/num_samples": torch.tensor(num_samples), "in/num_importance": torch.tensor(num_importance),
                "in/steps": torch.tensor(steps), "in/base_variance": torch.tensor(base_var)}
        for i, r in enumerate(rand):
            blob[f"in/rand{i}"] = r
        for k, v in p.items():
            blob[f"param/{k}"] = v
        for k, v in out.items():
            blob[f"out/{k}"] = v.detach()
        for k, v in losses.items():
            blob[f"loss/{k}"] = v.detach()
        for k, v in ref_grads.items():
            blob[f"grad/{k}"] = v
        blob.update(steps_blob)
        path = os.path.join(HERE, f"neus_small_{mode}.npz")
        np.savez_compressed(path, **{k: v.numpy() for k, v in blob.items()})
        print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


class _RandByShape:
    """torch.rand replacement that serves preset tensors per requested shape and records the order they were used in."""

    def __init__(self, pools):
        self.pools = {k: list(v) for k, v in pools.items()}
        self.used = []
        self._orig = torch.rand

    def __enter__(self):
        def fake(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            t = self.pools[shape].pop(0)
            self.used.append(t)
            return t.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def main_volsdf():
    """VolSDF (models/volsdf.py), BASELINE config 1 flavour: pure-MLP SDF field (use_grid_feature=False: zero grid features,
    sdf_field.py:389-390), ErrorBoundedSampler (ray_samplers.py:581), density rendering.  Nothing of tiny-cuda-nn is involved,
    so this golden pins the oracle against the reference with no shim in the loop.  Writes tests/golden/volsdf_small_{train,eval}.npz."""
    ns = ref_harness.import_reference()
    cfg = small_cfg()
    cfg.field.use_grid_feature = False
    p = {k: v for k, v in perturbed_params(cfg).items() if not k.startswith("proposal_networks.")}
    p["laplace_density.beta"] = torch.full((1,), 0.02)
    n = 40
    origins, dirs, cam = O.synthetic_rays(n, seed=44)
    g = torch.Generator().manual_seed(11)
    image = torch.rand(n, 3, generator=g)
    ns_final, ns_eval, ns_extra = 16, 32, 8
    pools = {(n, ns_eval + 1): [torch.rand(n, ns_eval + 1, generator=g) for _ in range(8)],
             (n, ns_final + 1): [torch.rand(n, ns_final + 1, generator=g) for _ in range(2)],
             (n, ns_extra + 1): [torch.rand(n, ns_extra + 1, generator=g) for _ in range(2)]}
    fc = cfg.field
    rcfg = ns.sf.SDFFieldConfig(
        num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim, num_layers_color=fc.num_layers_color,
        hidden_dim_color=fc.hidden_dim_color, bias=fc.bias, inside_outside=fc.inside_outside, use_grid_feature=False,
        beta_init=fc.beta_init, num_levels=fc.num_levels, max_res=fc.max_res, base_res=fc.base_res,
        log2_hashmap_size=fc.log2_hashmap_size, hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep)
    field = ns.sf.SDFField(rcfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=49,
                           spatial_distortion=ns.sd.SceneContraction(order=float("inf")))
    sd = field.state_dict()
    for k in sd:
        if k in p:
            sd[k] = p[k].clone()
    field.load_state_dict(sd)
    sampler = ns.rs.ErrorBoundedSampler(num_samples=ns_final, num_samples_eval=ns_eval, num_samples_extra=ns_extra)
    H = ns.FieldHeadNames
    for mode in ("train", "eval"):
        training = mode == "train"
        for m in [field, sampler]:
            m.train(training)
        rb = ns.rays.RayBundle(
            origins=origins, directions=dirs, pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
            camera_indices=cam[:, None], nears=torch.full((n, 1), cfg.near), fars=torch.full((n, 1), cfg.far))
        with _RandByShape(pools) as rq:
            ray_samples, _eik = sampler(rb, density_fn=field.laplace_density, sdf_fn=field.get_sdf)
        rand = list(rq.used) if training else None
        fo = field(ray_samples)
        weights, trans = ray_samples.get_weights_and_transmittance(fo[H.DENSITY])
        rgb_r = ns.rd.RGBRenderer(background_color=torch.zeros(3))
        rgb_r.train(training)
        rgb = rgb_r(rgb=fo[H.RGB], weights=weights)
        depth = ns.rd.DepthRenderer(method="expected")(weights=weights, ray_samples=ray_samples)
        normal = ns.rd.SemanticRenderer()(semantics=fo[H.NORMAL], weights=weights)
        acc = ns.rd.AccumulationRenderer()(weights=weights)
        out = {
            "starts": ray_samples.frustums.starts[..., 0], "ends": ray_samples.frustums.ends[..., 0],
            "bins": torch.cat([ray_samples.spacing_starts[..., 0], ray_samples.spacing_ends[..., -1:, 0]], -1),
            "sdf": fo[H.SDF][..., 0], "gradient": fo[H.GRADIENT], "field_rgb": fo[H.RGB], "density": fo[H.DENSITY][..., 0],
            "weights": weights[..., 0], "rgb": rgb, "depth": depth[..., 0], "normal": normal, "accumulation": acc[..., 0],
        }
        # ---- step-by-step replay with the reference's own methods (ray_samplers.py:633-682), recording every iteration
        steps_blob = {}
        sb = lambda r: torch.cat([r.spacing_starts[..., 0], r.spacing_ends[..., -1:, 0]], -1)
        with _RandQueue([t.clone() for t in rand] if training else []), torch.no_grad():
            beta0 = field.laplace_density.get_beta().detach()
            rs_k = sampler.uniform_sampler(rb, num_samples=ns_eval)
            deltas = rs_k.deltas.squeeze(-1)
            beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(sampler.eps + 1.0)))) * (deltas ** 2.0).sum(-1))
            total, not_conv, idx_k, new_k, sdf_k = 0, True, None, rs_k, None
            while not_conv and total < sampler.max_total_iters:
                new_sdf = field.get_sdf(new_k)
                sdf_k = new_sdf if idx_k is None else torch.gather(torch.cat([sdf_k.squeeze(-1), new_sdf.squeeze(-1)], -1), 1,
                                                                     idx_k).unsqueeze(-1)
                pre = f"step{total}/"
                steps_blob[pre + "bins_in"] = sb(rs_k)
                steps_blob[pre + "sdf_in"] = sdf_k.reshape(rs_k.shape).clone()
                steps_blob[pre + "beta_in"] = beta.clone()
                d_star = sampler.get_dstar(sdf_k, rs_k)
                beta = sampler.get_updated_beta(beta0, beta, field.laplace_density, sdf_k, d_star, rs_k)
                density = field.laplace_density(sdf_k.reshape(rs_k.shape), beta=beta.unsqueeze(-1))
                w, tr = rs_k.get_weights_and_transmittance(density.unsqueeze(-1))
                steps_blob[pre + "d_star"] = d_star.clone()
                steps_blob[pre + "beta_out"] = beta.clone()
                steps_blob[pre + "weights"] = w[..., 0].clone()
                total += 1
                not_conv = bool(beta.max() > beta0)
                if not_conv and total < sampler.max_total_iters:
                    deltas = rs_k.deltas.squeeze(-1)
                    eps_sec = torch.exp(-d_star / beta.unsqueeze(-1)) * (deltas ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
                    ew = (torch.clamp(torch.exp(torch.cumsum(eps_sec, dim=-1)), max=1.0e6) - 1.0) * tr[..., 0]
                    new_k = sampler.pdf_sampler(rb, rs_k, ew.unsqueeze(-1), num_samples=ns_eval)
                    steps_blob[pre + "err_weights"] = ew.clone()
                    steps_blob[pre + "new_bins"] = sb(new_k)
                    rs_k, idx_k = sampler.merge_ray_samples(rb, rs_k, new_k)
                    steps_blob[pre + "merged_bins"] = sb(rs_k)
                    steps_blob[pre + "index"] = idx_k.clone()
                else:
                    rs_k = sampler.pdf_sampler(rb, rs_k, w, num_samples=ns_final)
                    steps_blob[pre + "final_bins"] = sb(rs_k)
            n_iters = total
            uni = sampler.uniform_sampler(rb, num_samples=ns_extra)
            steps_blob["extra_bins"] = sb(uni)
            rs_k, _ = sampler.merge_ray_samples(rb, rs_k, uni)
            assert torch.equal(sb(rs_k), out["bins"]), "step-by-step replay differs from ErrorBoundedSampler.generate_ray_samples"
        print(f"[volsdf {mode}] reference sampler: {n_iters} outer iterations, beta0 {float(beta0):.3f}, final beta max "
              f"{float(steps_blob[f'step{n_iters - 1}/beta_out'].max()):.3f}")
        # ---- oracle: (1) whole sampler, (2) every iteration on the reference's inputs, (3) field + render on identical samples
        po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
        tr_o = []
        with torch.no_grad():
            o_s = O.volsdf_forward(origins, dirs, cam, p, cfg, rand=[t.clone() for t in rand] if training else None,
                                   training=training, num_samples=ns_final, num_samples_eval=ns_eval, num_samples_extra=ns_extra,
                                   trace=tr_o)
        d_bins = (o_s["bins"] - out["bins"]).abs()
        print(f"[volsdf {mode}] oracle sampler: {len(tr_o)} iterations, median |d bins| {d_bins.median().item():.1e}, max {d_bins.max().item():.1e}")
        assert len(tr_o) == n_iters and d_bins.median().item() <= 2e-6 and d_bins.max().item() <= 5e-3
        nears_t, fars_t = torch.full((n,), cfg.near), torch.full((n,), cfg.far)
        beta0_o = (p["laplace_density.beta"].abs() + p["laplace_density.beta_min"])
        for it in range(n_iters):
            pre = f"step{it}/"
            b_in, s_in, be_in = steps_blob[pre + "bins_in"], steps_blob[pre + "sdf_in"], steps_blob[pre + "beta_in"]
            eu = O.uniform_to_euclidean(b_in, nears_t, fars_t)
            dl = eu[:, 1:] - eu[:, :-1]
            ds = O.volsdf_dstar(s_in, dl)
            assert (ds - steps_blob[pre + "d_star"]).abs().max().item() <= 1e-6, f"step {it} d_star"
            be = O.volsdf_update_beta(beta0_o, be_in, s_in, ds, dl)
            assert (be - steps_blob[pre + "beta_out"]).abs().max().item() <= 1e-6 * float(be.max()), f"step {it} beta"
            w_o, t_o = O.weights_and_transmittance_from_density(O.laplace_density(s_in, be[:, None]), dl)
            assert (w_o - steps_blob[pre + "weights"]).abs().max().item() <= 2e-6, f"step {it} weights"
        losses, ref_grads = {}, {}
        if training:
            losses["rgb_loss"] = torch.nn.L1Loss()(image, rgb)
            losses["eikonal_loss"] = ((fo[H.GRADIENT].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
            field.zero_grad()
            sum(losses.values()).backward()
            for k, v in field.named_parameters():
                if v.grad is not None:
                    ref_grads[k] = v.grad.clone()
        o = O.volsdf_forward(origins, dirs, cam, po, cfg, training=training, samples=(out["bins"], out["starts"], out["ends"]))
        if not training:
            o["rgb"] = o["rgb"].clamp(0.0, 1.0)
        omap = {"sdf": o["field"]["sdf"], "gradient": o["field"]["gradient"], "field_rgb": o["field"]["rgb"],
                "density": o["field"]["density"], "weights": o["weights"], "rgb": o["rgb"], "depth": o["depth"],
                "normal": o["normal"], "accumulation": o["accumulation"]}
        worst = 0.0
        for k, v in omap.items():
            err = (v.detach() - out[k].detach()).abs().max().item()
            scale = out[k].detach().abs().max().item() + 1e-12
            worst = max(worst, err / scale)
            tol = 1e-4 if k == "depth" else 2e-5
            assert err <= tol * scale + 1e-6, f"[volsdf] oracle != reference on {k}: abs {err:.3e} (scale {scale:.3e})"
        if training:
            g_o = o["field"]["gradient"]
            ol = {"rgb_loss": torch.nn.functional.l1_loss(o["rgb"], image),
                  "eikonal_loss": ((g_o.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult}
            for k in losses:
                assert abs(ol[k].item() - losses[k].item()) <= 1e-5 * abs(losses[k].item()) + 1e-8, k
            sum(ol.values()).backward()
            for k, gref in ref_grads.items():
                err = (po[k].grad - gref).abs().max().item()
                scale = gref.abs().max().item() + 1e-12
                worst = max(worst, err / scale)
                assert err <= 1e-3 * scale + 1e-9, f"[volsdf] oracle grad != reference on {k}: {err:.3e} / {scale:.3e}"
        print(f"[volsdf {mode}] oracle reproduces the reference; worst rel err {worst:.2e}")
        blob = {"in/origins": origins, "in/dirs": dirs, "in/cam": cam, "in/image": image, "in/n_iters": torch.tensor(n_iters),
                "in/num_samples": torch.tensor(ns_final), "in/num_samples_eval": torch.tensor(ns_eval),
                "in/num_samples_extra": torch.tensor(ns_extra)}
        for i, r in enumerate(rand or []):
            blob[f"in/rand{i}"] = r
        for k, v in p.items():
            blob[f"param/{k}"] = v
        for k, v in out.items():
            blob[f"out/{k}"] = v.detach()
        for k, v in losses.items():
            blob[f"loss/{k}"] = v.detach()
        for k, v in ref_grads.items():
            blob[f"grad/{k}"] = v
        blob.update(steps_blob)
        path = os.path.join(HERE, f"volsdf_small_{mode}.npz")
        np.savez_compressed(path, **{k: v.numpy() for k, v in blob.items()})
        print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def main_numgrad():
    """use_numerical_gradients (the neus-facto-angelo field mode, sdf_field.py:431-453,638-644) on the small configuration:
    the reference SDFField evaluated on fixed ray samples, its finite-difference normals, `sampled_sdf`, the colour network
    on those normals, and the gradients of rgb-L1 + eikonal + curvature (neus_facto.py:312-325) w.r.t. every field parameter.
    Writes tests/golden/numgrad_small_train.npz."""
    ns = ref_harness.import_reference()
    cfg = small_cfg()
    p = {k: v for k, v in perturbed_params(cfg).items() if not k.startswith("proposal_networks.")}
    n, s = 40, 24
    delta = 1.0 / 128.0  # a coarse-level step as the delta schedule sets it early in training (neus_facto.py:209-225)
    origins, dirs, cam = O.synthetic_rays(n, seed=44)
    g = torch.Generator().manual_seed(11)
    starts = torch.sort(torch.rand(n, s, generator=g) * (cfg.far - cfg.near) + cfg.near, dim=-1).values
    ends = torch.cat([starts[:, 1:], torch.full((n, 1), cfg.far)], dim=-1)
    image = torch.rand(n, s, 3, generator=g)
    curv_mult = 5e-4
    field, _, _ = build_reference(ns, O.ModelCfg(field=cfg.field, proposals=()), p)
    field.config.use_numerical_gradients = True
    field.set_numerical_gradients_delta(delta)
    field.train(True)
    H = ns.FieldHeadNames
    rb = ns.rays.RayBundle(origins=origins, directions=dirs, pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
                           camera_indices=cam[:, None], nears=torch.full((n, 1), cfg.near), fars=torch.full((n, 1), cfg.far))
    rs = rb.get_ray_samples(bin_starts=starts[..., None], bin_ends=ends[..., None])
    fo = field(rs)
    sur = fo["sampled_sdf"].reshape(n, s, 3, 2)
    curvature = (sur.sum(dim=-1) - 2 * fo[H.SDF]) / (delta * delta)
    losses = {"rgb_loss": torch.nn.L1Loss()(image, fo[H.RGB]),
              "eikonal_loss": ((fo[H.GRADIENT].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult,
              "curvature_loss": torch.abs(curvature).mean() * curv_mult}
    field.zero_grad()
    sum(losses.values()).backward()
    ref_grads = {k: v.grad.clone() for k, v in field.named_parameters() if v.grad is not None}
    out = {"sdf": fo[H.SDF][..., 0], "gradient": fo[H.GRADIENT], "field_rgb": fo[H.RGB], "sampled_sdf": fo["sampled_sdf"],
           "normal": fo[H.NORMAL]}
    # ---- oracle on the same inputs
    po = {k: v.clone().requires_grad_(v.is_floating_point() and k != "laplace_density.beta_min") for k, v in p.items()}
    o = O.field_outputs(origins, dirs, starts, ends - starts, cam, po, cfg.field, None, 1.0, True, numerical_delta=delta)
    omap = {"sdf": o["sdf"], "gradient": o["gradient"], "field_rgb": o["rgb"], "sampled_sdf": o["sampled_sdf"], "normal": o["normal"]}
    worst = 0.0
    for k, v in out.items():
        err = (omap[k].detach() - v.detach()).abs().max().item()
        scale = v.detach().abs().max().item() + 1e-12
        worst = max(worst, err / scale)
        # the finite difference divides fp32 round-off of the sdf (1e-7) by 2 delta
        tol = 1e-4 if k in ("gradient", "normal") else 2e-5
        assert err <= tol * scale + 1e-6, f"[numgrad] oracle != reference on {k}: abs {err:.3e} (scale {scale:.3e})"
    ol = {"rgb_loss": torch.nn.functional.l1_loss(o["rgb"], image),
          "eikonal_loss": ((o["gradient"].norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult,
          "curvature_loss": ((o["sampled_sdf"].reshape(n, s, 3, 2).sum(-1) - 2 * o["sdf"][..., None]) / (delta * delta)).abs().mean() * curv_mult}
    for k in losses:
        assert abs(ol[k].item() - losses[k].item()) <= 2e-4 * abs(losses[k].item()) + 1e-8, (k, ol[k].item(), losses[k].item())
    sum(ol.values()).backward()
    for k, gref in ref_grads.items():
        err = (po[k].grad - gref).abs().max().item()
        scale = gref.abs().max().item() + 1e-12
        worst = max(worst, err / scale)
        assert err <= 2e-3 * scale + 1e-9, f"[numgrad] oracle grad != reference on {k}: {err:.3e} / {scale:.3e}"
    print(f"[numgrad] oracle reproduces the reference; worst rel err {worst:.2e}")
    blob = {"in/origins": origins, "in/dirs": dirs, "in/cam": cam, "in/starts": starts, "in/ends": ends, "in/image": image,
            "in/delta": torch.tensor(delta), "in/curv_mult": torch.tensor(curv_mult)}
    for k, v in p.items():
        blob[f"param/{k}"] = v
    for k, v in out.items():
        blob[f"out/{k}"] = v.detach()
    for k, v in losses.items():
        blob[f"loss/{k}"] = v.detach()
    for k, v in ref_grads.items():
        blob[f"grad/{k}"] = v
    path = os.path.join(HERE, "numgrad_small_train.npz")
    np.savez_compressed(path, **{k: v.detach().numpy() for k, v in blob.items()})
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "facto"):
        main()
    if which in ("all", "neus"):
        main_neus()
    if which in ("all", "volsdf"):
        main_volsdf()
    if which in ("all", "numgrad"):
        main_numgrad()
