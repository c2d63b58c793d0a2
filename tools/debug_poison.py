"""Debug aid: does a training step of the golden NeuS-facto model depend on what is in freed GPU memory?  Runs the step under different
fillings of the caching allocator's free blocks and reports every output / gradient that is not bit-identical to the first run."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, product_model_from_params, small_oracle_cfg  # noqa: E402
from test_gpu_parity import _bundle  # noqa: E402

dev = torch.device("cuda:0")
g = load_golden("train")
cfg = small_oracle_cfg()


def poison(value):
    blocks = []
    try:
        for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16):
            for _ in range(3):
                blocks.append(torch.full((n,), value, device="cuda"))
    except RuntimeError:
        pass
    del blocks


def step():
    model = product_model_from_params(g["param"], cfg, dev).train()
    torch.manual_seed(3)
    rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, dev)
    out = model(rb)
    losses = model.get_loss_dict(out, {"image": g["in"]["image"]})
    loss = sum(losses.values())
    model.zero_grad()
    loss.backward()
    res = {f"out.{k}": v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
    res.update({f"loss.{k}": v.detach().clone() for k, v in losses.items()})
    res.update({f"grad.{k}": p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    return res


from sdfstudio_amd.distributed import FlatGradients  # noqa: E402


def slotted_steps(n):
    model = product_model_from_params(g["param"], cfg, dev).train()
    groups = {k: v for k, v in model.get_param_groups().items() if v}
    flat = FlatGradients([p for grp in groups.values() for p in grp], buckets=list(groups.values()))
    outs = []
    for i in range(n):
        torch.manual_seed(3)
        rb = _bundle(g["in"]["origins"], g["in"]["dirs"], g["in"]["cam"], cfg.near, cfg.far, dev)
        out = model(rb)
        loss = sum(model.get_loss_dict(out, {"image": g["in"]["image"]}).values())
        flat.zero()
        loss.backward()
        flat.finish()
        outs.append({f"grad.{k}": p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    return outs


def report(tag, r, ref):
    bad = []
    for k in ref:
        if k in r and not torch.equal(r[k], ref[k]):
            d = (r[k].float() - ref[k].float()).abs()
            bad.append((k, float(d.max()), float(ref[k].float().abs().max()), int((d > 0).sum()), d.numel()))
    print(f"== {tag}: {len(bad)} tensors differ")
    for b in bad:
        print("   %-60s max|d| %.3e (scale %.3e) %d / %d elements" % b)


poison(0.0)
plain = step()
for name, val in (("zeros", 0.0), ("nan", float("nan")), ("1e30", 1e30)):
    poison(val)
    for i, r in enumerate(slotted_steps(2)):
        report(f"slotted, poison {name}, pass {i} vs plain", r, plain)
sys.exit(0)

ref = None
for name, val in (("zeros", 0.0), ("nan", float("nan")), ("1e30", 1e30), ("zeros again", 0.0), ("-3.7", -3.7)):
    poison(val)
    r = step()
    if ref is None:
        ref = r
        print("reference run:", name, len(r), "tensors")
        continue
    bad = []
    for k in ref:
        if not torch.equal(r[k], ref[k]):
            d = (r[k].float() - ref[k].float()).abs()
            bad.append((k, float(d.max()), float(ref[k].float().abs().max()), int((d > 0).sum()), d.numel()))
    print(f"== {name}: {len(bad)} tensors differ")
    for b in bad:
        print("   %-60s max|d| %.3e (scale %.3e) %d / %d elements" % b)
