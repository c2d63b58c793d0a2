"""The data-parallel exchange EXECUTED ON RCCL (-m gpu).  Rows a19 / (e) / g3 of SURVEY section 8 had only ever run over
gloo's CPU collectives.  A `gpurun` box has one MI355X, so the process group here has ONE rank - but with backend "nccl" (= RCCL) and
SDFHIP_FORCE_EXCHANGE=1 every reduce_scatter_tensor / all_gather_into_tensor / all_reduce the N > 1 path issues IS issued: RCCL kernels on
RCCL's stream, stream waits on the compute stream, the side-stream parameter gathers with wait_parameters(late=...), and the native
sdfhip table-gradient callback -> ExternalStream -> launch ordering.  Each collective of one rank returns its input, so the run must be the
same training as with no exchange at all.  What replaces: pipelines/base_pipeline.py:241-243 (DDP), scripts/train.py:127-145."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(case, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDFHIP_FORCE_EXCHANGE", "SDFHIP_BENCH_EXCHANGE", "SDFHIP_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_worker.py"), case], env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_rccl_protocol_bit_identical_to_no_exchange():
    """Deterministic gradients, the fused Adam kernel, chunked buckets, a late bucket, an unused parameter, a level switched on at step 2:
    parameters and both Adam moments after 5 steps are THE SAME BITS under reduce-scatter / all-gather, under all-reduce and with no
    exchange."""
    rep = _worker("protocol")
    assert rep["backend"] == "nccl"
    assert rep["params_equal"] == {"shard": True, "allreduce": True}, rep
    assert rep["moments_equal"] == {"shard": True, "allreduce": True}, rep
    assert rep["table_moved"] > 0.99 and rep["table_tail_untouched"]  # rows of the levels never switched on: neither exchanged nor stepped
    assert all(c == [0, 0] for c in rep["collectives"]["none"]), rep["collectives"]
    # the table is 41 000 elements in chunks of 4096: the active prefix 16 000 -> 4 chunks, 29 000 -> 8; + 1 bucket each for the rest of
    # "fields" and for "proposal_networks"
    sh = rep["collectives"]["shard"]
    assert sh[0][0] == 4 + 2 and sh[-1][0] == 8 + 2, sh
    assert sh[-1][1] > sh[0][1] > 0, sh  # parameter all-gathers were issued (cumulative counter)
    ar = rep["collectives"]["allreduce"]
    assert ar[0][0] >= 2 and ar[0][1] == 0, ar


@pytest.mark.parametrize("case", ["small", "config5"])
def test_rccl_real_training_steps_match_no_exchange(case):
    """The real step (bench.make_job: sample -> field -> render -> losses -> backward -> exchange -> fused Adam), small parity configuration
    and BASELINE config 5 at its own size ACROSS a progressive-level switch (8 -> 9 at step 80 000), under both exchanges on RCCL."""
    rep = _worker(case, timeout=1500)
    assert rep["backend"] == "nccl"
    info = rep["info"]
    assert not info["none"]["exchanging"] and info["shard"]["exchanging"] and info["allreduce"]["exchanging"]
    assert info["shard"]["shard"] and not info["allreduce"]["shard"]
    assert info["shard"]["collectives_last_step"] >= 2 and info["allreduce"]["collectives_last_step"] >= 2
    assert info["shard"]["gather_collectives_total"] >= 2
    assert info["shard"]["early_buckets_last_step"] == 1, info  # the SDF table left from INSIDE the field's backward (native callback)
    assert info["shard"]["overlapped_buckets_last_step"] >= 1 and info["allreduce"]["overlapped_buckets_last_step"] >= 1
    if case == "config5":
        assert info["none"]["levels"] == [8, 8, 9, 9] == info["shard"]["levels"] == info["allreduce"]["levels"], info
    # the MLP weights come out of kernels with a fixed summation order: the two plain runs agree on them bit for bit, and so must the
    # runs with RCCL in the loop
    assert rep["deterministic_after_step1"] >= 10, rep
    for mode in ("shard", "allreduce"):
        m = rep["modes"][mode]
        assert m["all_finite"], (mode, m)
        assert m["bit_identical_on_deterministic_params_after_step1"], (mode, m)
        assert m["end_within_plain_runs_spread"], (mode, m)
