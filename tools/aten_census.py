"""Which ATen operators does one training step still run, and from which line of this package?  A TorchDispatchMode over 3 steps: every
operator that reaches the dispatcher with a device tensor (views and metadata queries excluded), grouped by the innermost frame of
sdfstudio_amd / bench.py on the Python stack (operators run by the autograd engine's own nodes have no Python frame: "autograd engine").
    python tools/aten_census.py [config 2|5] [first_step]   ->  table on stdout"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VIEWS = ("view", "reshape", "expand", "slice", "select", "unsqueeze", "squeeze", "transpose", "permute", "alias", "detach", "t.default", "as_strided",
         "unbind", "split", "chunk", "narrow", "_unsafe_view", "size", "stride", "numel", "is_", "empty", "_local_scalar_dense", "item", "sym_", "lift_fresh",
         "unflatten", "flatten", "movedim", "broadcast_to", "new_empty", "result_type", "_has_", "to.dtype_layout", "record_stream")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if not any(v in name for v in VIEWS):
            where = "autograd engine"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if (ROOT in fr.filename) and "tools/aten_census" not in fr.filename:
                    where = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.name}"
                    break
            self.rows[(where, name)] += 1
        return out


config = int(sys.argv[1]) if len(sys.argv) > 1 else 5
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
eval_mode = len(sys.argv) > 3 and sys.argv[3] == "eval"  # the forward-only leg of bench.py: eval-mode render of one batch, no grad
dev = torch.device("cuda:0")
if config == 5:
    bench.N_RAYS, bench.N_SAMPLES = 2048, 48
job = bench.make_job(config, dev, 1, 0)
for i in range(4):
    job["step"](first + i)
torch.cuda.synchronize()
N = 3
census = Census()
if eval_mode:
    import time

    from sdfstudio_amd.cameras.rays import RayBundle

    model = job["model"].eval()
    o, d, norm, cam = bench.draw_rays(job["centers"], job["rot"], job["n_rays"], job["gen"])
    rb = RayBundle(origins=o, directions=d, directions_norm=norm, camera_indices=cam[:, None])
    with torch.no_grad():
        model(rb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(rb)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"eval batch: host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms, total {(t2 - t0) / 20 * 1e3:.3f} ms")
        with census:
            for i in range(N):
                model(rb)
else:
    with census:
        for i in range(N):
            job["step"](first + 4 + i)
torch.cuda.synchronize()
tot = 0
for (where, op), n in sorted(census.rows.items(), key=lambda kv: (kv[0][0], -kv[1])):
    tot += n
    print(f"{n / N:6.1f}/step  {where[:100]:100s} {op}")
print(f"TOTAL operators per step: {tot / N:.1f}")
