"""Debug aid: is the no-grad field forward of the golden model bit-reproducible across calls / allocator states?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, product_model_from_params, small_oracle_cfg  # noqa: E402
from test_gpu_parity import _bundle  # noqa: E402

dev = torch.device("cuda:0")


def poison(value):
    blocks = []
    try:
        for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16):
            for _ in range(3):
                blocks.append(torch.full((n,), value, device="cuda"))
    except RuntimeError:
        pass
    del blocks


for mode in ("train", "eval"):
    g = load_golden(mode)
    cfg = small_oracle_cfg()
    ref = None
    for it, val in enumerate((0.0, float("nan"), 1e30, -3.7, 0.0, float("nan"), 1e30, 5e-41, float("inf"), 1e30)):
        poison(val)
        model = product_model_from_params(g["param"], cfg, dev).train(mode == "train")
        model.field.set_cos_anneal_ratio(float(g["in"]["cos_anneal"]))
        rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, dev)
        rs = rb.get_ray_samples(g["out"]["starts"].to(dev), g["out"]["ends"].to(dev))
        outs = []
        for rep in range(3):
            with torch.no_grad():
                sdf, grad, rgb, x = model.field.forward_fused(rs)
            outs.append((sdf.clone(), grad.clone(), rgb.clone()))
        if ref is None:
            ref = outs[0]
            variants = [outs[0][2]]
        line = []
        for rep, o in enumerate(outs):
            for vi, v in enumerate(variants):
                if torch.equal(o[2], v):
                    break
            else:
                variants.append(o[2])
                vi = len(variants) - 1
            line.append(vi)
            assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]), "sdf / grad differ"
        print(f"{mode} poison#{it}={val}: rgb variant per call {line}")
    gref = g["out"]["field_rgb"].to(dev)
    for vi, v in enumerate(variants):
        d = (v.reshape(gref.shape) - gref).abs()
        pts = (v.reshape(-1, 3) != variants[0].reshape(-1, 3)).any(dim=1).nonzero().flatten()
        print(f"{mode} variant {vi}: max|rgb - golden| {float(d.max()):.3e}; points differing from variant 0: {pts.numel()}",
              (f"first {int(pts[0])} last {int(pts[-1])}" if pts.numel() else ""))
