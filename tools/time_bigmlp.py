"""Step time of the neus-facto-bigmlp field shape (configs/method_configs.py:503-523: SDFFieldConfig(num_layers=8, hidden_dim=512,
num_layers_color=4), 2048 rays per batch, the model's default 48 field samples + 256 / 96 proposal samples) next to the same step at
hidden 256 - the 512-wide geometry network runs layer by layer (csrc/wide_kernels.h).  Not a BASELINE config: a documentation number.
    python tools/time_bigmlp.py [steps] [HIDDENxRAYSxSAMPLES]   ->  one JSON line per case"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(hidden, rays, samples, steps):
    from sdfstudio_amd.cameras.rays import RayBundle
    from sdfstudio_amd.distributed import FlatGradients
    from sdfstudio_amd.engine.optimizers import Optimizers, multi_step_scheduler, neus_scheduler
    from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
    from sdfstudio_amd.models.neus_facto import NeuSFactoModel, NeuSFactoModelConfig, SceneBox

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    fcfg = SDFFieldConfig(num_layers=8, hidden_dim=hidden, num_layers_color=4, bias=0.5, inside_outside=False, beta_init=0.3)
    mcfg = NeuSFactoModelConfig(sdf_field=fcfg, num_neus_samples_per_ray=samples, background_model="none")
    model = NeuSFactoModel(mcfg, SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=0.5, far=4.5), num_train_data=49).to(dev).train()
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    flat = FlatGradients([p for g in groups.values() for p in g], buckets=list(groups.values()))
    opts = Optimizers({"fields": {"lr": 5e-4, "scheduler": neus_scheduler(500, 0.05, 20000)},
                       "proposal_networks": {"lr": 1e-2, "scheduler": multi_step_scheduler(20000)}}, groups, flat_grads=flat)
    centers, rot = bench.synthetic_cameras(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)

    def step(i):
        model.before_train_iteration(i)
        o, d, norm, cam = bench.draw_rays(centers, rot, rays, gen)
        image = torch.rand(rays, 3, device=dev, generator=gen)
        out = model(RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None]))
        loss = sum(model.get_loss_dict(out, {"image": image}).values())
        flat.zero()
        loss.backward()
        opts.optimizer_step_all(grad_scale=flat.finish(average=False))
        opts.scheduler_step_all(i)
        return loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(3 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert torch.isfinite(loss)
    print(json.dumps({"hidden_dim": hidden, "rays": rays, "samples_per_ray": samples, "ms_per_step": round(ms, 3),
                      "ray_samples_per_s": round(rays * samples / ms * 1e3, 1)}))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    if len(sys.argv) > 2:  # one case only (profiling):  time_bigmlp.py 5 512x4096x128
        h, r, s_ = (int(v) for v in sys.argv[2].split("x"))
        run(h, r, s_, n)
    else:
        for h in (256, 512):
            run(h, 2048, 48, n)
        run(512, 4096, 128, n)
