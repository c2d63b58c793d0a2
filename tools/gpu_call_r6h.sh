#!/bin/bash
# GPU call H (round 6): ref-nerf / off-axis / periodic field tests, per-field hooks, RCCL small case (criterion pooled), mesh with 7 goldens.
mkdir -p gpurun_out/r6h
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_gpu_refnerf.py tests/test_gpu_field_hooks.py tests/test_gpu_rccl_single_rank.py tests/test_gpu_zy_mesh_abi.py tests/test_gpu_zz_mesh.py -q -m gpu > gpurun_out/r6h/pytest.log 2>&1
echo "pytest rc=$?"; grep -v Warning gpurun_out/r6h/pytest.log | tail -40
timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6h/mesh_gpu_check.jsonl; echo "mesh check rc=$?"; grep -c '"ok": 1' gpurun_out/r6h/mesh_gpu_check.jsonl; tail -2 gpurun_out/r6h/mesh_gpu_check.jsonl
