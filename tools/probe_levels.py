"""Dev probe (GPU box): time grid_bwd_kernel / prop_bwd_kernel per hash level via the level mask + sdfhip_profile_*."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sdfstudio_amd import _lib
from sdfstudio_amd.cameras.rays import RayBundle

dev = torch.device("cuda:0")
model = bench.build_model(dev)
centers, rot = bench.synthetic_cameras(dev)
gen = torch.Generator(device=dev); gen.manual_seed(42)
o, d, norm, cam = bench.draw_rays(centers, rot, 4096, gen)
image = torch.rand(4096, 3, device=dev, generator=gen)

def run(mask_levels):
    m = torch.zeros(32, device=dev)
    for l in mask_levels:
        m[2 * l:2 * l + 2] = 1
    model.field.hash_encoding_mask = m
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize(); _lib.profile_enable(True)
        rb = RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None])
        out = model(rb)
        loss = sum(model.get_loss_dict(out, {"image": image}).values())
        model.zero_grad(); loss.backward()
    torch.cuda.synchronize()
    p = _lib.profile_collect(); _lib.profile_enable(False)
    return {k: round(v[0] / 2, 3) for k, v in p.items() if k in ("grid_bwd_kernel", "geo_encode_kernel", "prop_bwd_kernel")}

res = {"all": run(range(16))}
for l in range(16):
    res[f"level{l}"] = run([l])
print(json.dumps(res, indent=1))
