"""Mint golden vectors for the BACKGROUND-MODEL paths by running the reference's own model classes (build container only).

    python tests/golden/make_golden_bg.py     # writes tests/golden/{neus,volsdf,neus_facto}_bg_mlp_eval.npz and
                                              # {neus,neus_facto}_bg_grid_eval.npz (GOLDEN_ONLY=grid: the latter two only)

Unlike make_golden.py (which drives field / sampler / renderer objects by hand), this script instantiates the reference's
``NeuSModel``, ``VolSDFModel`` and ``NeuSFactoModel`` (nerfstudio/models/*.py, unmodified, imported through
oracle/ref_harness.py with the documented PyTorch tinycudann shim) with ``background_model="mlp"`` - the reference's default
(base_surface_model.py:123) - and records, in eval mode (deterministic samplers): the ray inputs, the complete state_dict,
``model(ray_bundle)`` outputs, the rgb L1 loss and its gradient w.r.t. every parameter.  Covers
  * neus.py:94-104 + base_surface_model.py:314-329   (transmittance x colour of the samples beyond the far plane),
  * volsdf.py:62-79 + the same,
  * neus_facto.py:286-292 + base_surface_model.py:266-290 (background merged into alpha / colour outside the unit sphere).
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

FIELD = dict(num_layers=8, hidden_dim=64, geo_feat_dim=64, num_layers_color=4, hidden_dim_color=64, bias=0.5, inside_outside=False,
             use_grid_feature=True, beta_init=0.3, num_levels=8, max_res=128, base_res=4, log2_hashmap_size=11,
             hash_features_per_level=2, hash_smoothstep=True)
PROPS = [{"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 32, "base_res": 4},
         {"hidden_dim": 16, "log2_hashmap_size": 9, "num_levels": 5, "max_res": 64, "base_res": 4}]


def perturb(model, seed):
    """Noise on every parameter so that no path is dead (geometric init zeroes most first-layer columns)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if not p.requires_grad or p.numel() == 1:
                continue
            if "encoding.params" in k or k.endswith("mlp_base.params"):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.3)
            elif k.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.add_(0.05 * torch.randn(p.shape, generator=g))


def main():
    ns = ref_harness.import_reference()
    import nerfstudio.models.neus as rn
    import nerfstudio.models.neus_facto as rnf
    import nerfstudio.models.volsdf as rv
    from nerfstudio.data.scene_box import SceneBox

    sb = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5, radius=1.0, collider_type="near_far")
    fcfg = ns.sf.SDFFieldConfig(**FIELD)
    n = 48
    o, d, cam = O.synthetic_rays(n, seed=11)
    g = torch.Generator().manual_seed(3)
    image = torch.rand(n, 3, generator=g)
    # background_model="grid" (BASELINE config 5, method_configs.py:423): base_surface_model.py:181-187 builds TCNNNerfactoField with
    # its defaults (16 levels x 2^19 entries: a 67 MB table); the golden shrinks the table through the constructor's own arguments
    import functools

    import nerfstudio.fields.nerfacto_field as rnff
    import nerfstudio.models.base_surface_model as rbsm

    rbsm.TCNNNerfactoField = functools.partial(rnff.TCNNNerfactoField, num_levels=6, max_res=64, log2_hashmap_size=10)
    builds = {
        "neus": lambda: rn.NeuSModelConfig(sdf_field=fcfg, background_model="mlp", num_samples=16, num_samples_importance=16,
                                           num_up_sample_steps=2, num_samples_outside=8),
        "volsdf": lambda: rv.VolSDFModelConfig(sdf_field=fcfg, background_model="mlp", num_samples=16, num_samples_eval=32,
                                               num_samples_extra=8, num_samples_outside=8),
        "neus_facto": lambda: rnf.NeuSFactoModelConfig(sdf_field=fcfg, background_model="mlp", num_proposal_samples_per_ray=(32, 24),
                                                       num_neus_samples_per_ray=16, proposal_net_args_list=PROPS, num_samples_outside=8),
        "neus_facto_grid": lambda: rnf.NeuSFactoModelConfig(sdf_field=fcfg, background_model="grid", num_proposal_samples_per_ray=(32, 24),
                                                            num_neus_samples_per_ray=16, proposal_net_args_list=PROPS, num_samples_outside=8),
        "neus_grid": lambda: rn.NeuSModelConfig(sdf_field=fcfg, background_model="grid", num_samples=16, num_samples_importance=16,
                                                num_up_sample_steps=2, num_samples_outside=8),
    }
    only = os.environ.get("GOLDEN_ONLY")  # e.g. GOLDEN_ONLY=grid re-mints the two grid-background goldens only
    for name, mk in builds.items():
        if only == "grid" and not name.endswith("_grid"):
            continue
        torch.manual_seed(0)
        model = mk().setup(scene_box=sb, num_train_data=49, world_size=1, local_rank=0)
        perturb(model, seed=5)
        model.eval()
        rb = ns.rays.RayBundle(origins=o.clone(), directions=d.clone(), pixel_area=torch.ones(n, 1), directions_norm=torch.ones(n, 1),
                               camera_indices=cam[:, None])
        out = model(rb)
        loss = torch.nn.functional.l1_loss(image, out["rgb"])
        model.zero_grad()
        loss.backward()
        blob = {"in/origins": o.numpy(), "in/directions": d.numpy(), "in/camera_indices": cam.numpy(), "in/image": image.numpy(),
                "loss/rgb_loss": loss.detach().numpy()}
        for k in ("rgb", "depth", "normal", "accumulation", "weights"):
            blob[f"out/{k}"] = out[k].detach().numpy()
        for k, v in model.state_dict().items():
            blob[f"param/{k}"] = v.detach().numpy()
        n_grad = 0
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            # the 8 x 256 background MLP dominates the file: keep the gradients of its first, skip and last base layers, its head
            # MLP and its two output heads (every other gradient of the model is kept)
            if k.startswith("field_background.mlp_base.layers.") and int(k.split(".")[3]) not in (0, 4, 7):
                continue
            blob[f"grad/{k}"] = p.grad.detach().numpy()
            n_grad += 1
        bg_keys = [k for k in blob if k.startswith("grad/field_background")]
        assert bg_keys and all(np.abs(blob[k]).max() > 0 for k in bg_keys), "the background field must receive gradient"
        path = os.path.join(HERE, f"{name}_bg_mlp_eval.npz" if not name.endswith("_grid") else f"{name[:-5]}_bg_grid_eval.npz")
        np.savez_compressed(path, **blob)
        print(f"[{name}] rgb mean {out['rgb'].mean().item():.4f}, loss {loss.item():.5f}, {n_grad} parameter gradients -> {path} "
              f"({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
