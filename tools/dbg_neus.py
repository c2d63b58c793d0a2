"""Debug aid: colour-network gradients of the NeuS golden case under variations; dumps to gpurun_out/dbg_<tag>.pt"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import load_golden_file, small_oracle_cfg, load_params, product_grads
from test_gpu_parity import _bundle
from sdfstudio_amd.fields.sdf_field import SDFFieldConfig
from sdfstudio_amd.model_components.renderers import neus_render
from sdfstudio_amd.models.neus import NeuSModel, NeuSModelConfig
from sdfstudio_amd.models.neus_facto import SceneBox

tag, first, dense = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
device = torch.device("cuda:0")
g = load_golden_file("neus_small_train.npz")
cfg = small_oracle_cfg(); fc = cfg.field
fcfg = SDFFieldConfig(num_layers=fc.num_layers, hidden_dim=fc.hidden_dim, geo_feat_dim=fc.geo_feat_dim,
                      num_layers_color=fc.num_layers_color, hidden_dim_color=fc.hidden_dim_color, bias=fc.bias,
                      inside_outside=fc.inside_outside, use_grid_feature=True, beta_init=fc.beta_init, num_levels=fc.num_levels,
                      max_res=fc.max_res, base_res=fc.base_res, log2_hashmap_size=fc.log2_hashmap_size,
                      hash_features_per_level=fc.hash_features_per_level, hash_smoothstep=fc.hash_smoothstep)
steps = int(g["in"]["steps"])
mcfg = NeuSModelConfig(sdf_field=fcfg, num_samples=int(g["in"]["num_samples"]), num_samples_importance=int(g["in"]["num_importance"]),
                       num_up_sample_steps=steps, base_variance=float(g["in"]["base_variance"]), eikonal_loss_mult=cfg.eikonal_loss_mult)
box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), near=cfg.near, far=cfg.far)
model = NeuSModel(mcfg, box, num_train_data=49)
load_params(model, g["param"])
model = model.to(device).train(True)
ca = float(g["in"]["cos_anneal"]); model.field.set_cos_anneal_ratio(ca)
model.sampler.uniform_sampler.jitter_override = g["in"]["rand0"].to(device)
model.sampler.jitter_overrides = [g["in"][f"rand{1 + i}"].to(device) for i in range(steps)]
rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, device)
ref = g["out"]
if first:
    out = model(rb)
rs = rb.get_ray_samples(ref["starts"].to(device), ref["ends"].to(device))
sdf, grad, rgb, x = model.field.forward_fused(rs)
print("P =", sdf.numel(), "rgb nan", torch.isnan(rgb).any().item())
if dense == 2:  # per-tile contributions
    torch.manual_seed(3)
    R = torch.randn_like(rgb)
    rows = []
    nt = rgb.reshape(-1, 3).shape[0] // 32
    for t in range(nt):
        m = torch.zeros_like(rgb).reshape(-1, 3); m[t * 32:(t + 1) * 32] = 1
        model.zero_grad()
        ((rgb * R).reshape(-1, 3) * m).sum().backward(retain_graph=True)
        gg = product_grads(model)
        rows.append(torch.cat([gg["clin3.bias"].flatten().cpu(), gg["clin0.bias"].flatten().cpu()]))
    rows = torch.stack(rows)
    torch.save(rows, f"gpurun_out/dbg_{tag}.pt")
    if len(sys.argv) > 4:
        o = torch.load(f"gpurun_out/dbg_{sys.argv[4]}.pt")
        for t in range(nt):
            d = (rows[t] - o[t]).abs().max().item(); sc = o[t].abs().max().item()
            print(f"tile {t:3d} max|d| {d:.3e} scale {sc:.3e} rel {d / max(sc, 1e-30):.2e}")
    sys.exit(0)
if dense:
    torch.manual_seed(3)
    loss = (rgb * torch.randn_like(rgb)).sum()
else:
    out_rgb, depth, normal, acc, weights, alpha = neus_render(sdf, grad, rgb, model.field.deviation_network.variance, rs.flat_directions,
                                                              rs.flat_starts, rs.flat_ends, ca, None)
    loss = F.l1_loss(g["in"]["image"].to(device), out_rgb) + ((grad.norm(2, dim=-1) - 1) ** 2).mean() * cfg.eikonal_loss_mult
model.zero_grad(); loss.backward()
got = {k: v.detach().cpu() for k, v in product_grads(model).items()}
torch.save({"grads": got, "rgb": rgb.detach().cpu()}, f"gpurun_out/dbg_{tag}.pt")
if len(sys.argv) > 4:
    other = torch.load(f"gpurun_out/dbg_{sys.argv[4]}.pt")
    for k, v in got.items():
        o = other["grads"][k]
        d = (v - o).abs().max().item(); sc = o.abs().max().item()
        if k.startswith("clin") or k.startswith("enc") or d > 1e-3 * sc:
            print(f"{k:28s} max|d| {d:.3e} scale {sc:.3e} rel {d / max(sc, 1e-30):.2e}")
    print("rgb diff", (got and (torch.load(f'gpurun_out/dbg_{tag}.pt')['rgb'] - other['rgb']).abs().max().item()))
