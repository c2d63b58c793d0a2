"""Is a bench step bound by the GPU or by the host that enqueues it?  Times N steps twice: until the host has ENQUEUED the last one, and
until the GPU has finished it.  enqueue << total: the GPU is the limit and launch overhead is hidden behind it.
    python tools/enqueue_vs_gpu.py [config 2|5] [first_step]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

config = int(sys.argv[1]) if len(sys.argv) > 1 else 5
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
if config == 5:
    bench.N_RAYS, bench.N_SAMPLES = 2048, 48
job = bench.make_job(config, dev, 1, 0)
for i in range(5):
    job["step"](first + i)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    job["step"](first + 5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"config": config, "first_step": first, "host_enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 3),
                  "total_ms_per_step": round((t2 - t0) / n * 1e3, 3), "gpu_tail_ms_after_last_enqueue": round((t2 - t1) * 1e3, 3)}))
