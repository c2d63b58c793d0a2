"""The N > 1 control flow of bench.py on a single-GPU box: `python bench.py --gpus 2` must spawn its own ranks (the driver's
SCALE run launches it either way), run the REAL training step on the small configuration with the gradient exchange over
gloo (SDFHIP_BENCH_BACKEND=gloo: two ranks share cuda:0; RCCL needs one device per rank) and print one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_self_spawns_two_ranks():
    env = dict(os.environ, SDFHIP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--small"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["collective"]["backend"] == "gloo" and d["collective"]["bytes_per_step_per_rank"] > 0
    assert "rank 1/2" in r.stderr and "rank 0/2" in r.stderr
    # the exchange must OVERLAP with backward on the real model: laplace_density.beta and the (switched-off) appearance embedding are
    # outside a NeuS-facto step's graph and must not keep the "fields" bucket waiting for finish()
    assert d["collective"]["parameters_outside_the_graph"] >= 1, d["collective"]
    assert d["collective"]["buckets_launched_during_backward"] >= 1, d["collective"]
