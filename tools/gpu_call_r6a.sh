#!/bin/bash
# GPU call A (round 6): RCCL on one rank (tests + bench leg), the mesh library's "before" profile (tools/profile_mesh.sh, never run in r5).
mkdir -p gpurun_out/r6a
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1700 python -m pytest tests/test_gpu_rccl_single_rank.py -x -q -m gpu > gpurun_out/r6a/pytest_rccl.log 2>&1
echo "pytest rccl rc=$?"; tail -30 gpurun_out/r6a/pytest_rccl.log
timeout 600 python bench.py --only exchange > gpurun_out/r6a/exchange.json 2> gpurun_out/r6a/exchange.err
echo "exchange rc=$?"; cat gpurun_out/r6a/exchange.json; tail -5 gpurun_out/r6a/exchange.err
timeout 600 bash tools/profile_mesh.sh r6a/mesh_before > gpurun_out/r6a/profile_mesh.log 2>&1
echo "profile_mesh rc=$?"
cat gpurun_out/r6a/mesh_before/check_kt.jsonl | tail -3
