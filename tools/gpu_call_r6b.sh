#!/bin/bash
# GPU call B (round 6): the rewritten mesh library - C ABI check (goldens + crop 512 bit-exact, timing), rocprofv3 stats + PMC passes, the
# Python-binding tests (no xfail any more).
mkdir -p gpurun_out/r6b
timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6b/mesh_gpu_check.jsonl
echo "mesh_gpu_check rc=$?"
cat gpurun_out/r6b/mesh_gpu_check.jsonl
timeout 600 bash tools/profile_mesh.sh r6b/mesh > gpurun_out/r6b/profile_mesh.log 2>&1
echo "profile_mesh rc=$?"
timeout 1500 python -m pytest tests/test_gpu_zy_mesh_abi.py tests/test_gpu_zz_mesh.py -x -q -m gpu > gpurun_out/r6b/pytest_mesh.log 2>&1
echo "pytest mesh rc=$?"; tail -15 gpurun_out/r6b/pytest_mesh.log
timeout 300 python tools/mesh_leg.py > gpurun_out/r6b/mesh_leg.json 2> gpurun_out/r6b/mesh_leg.err
echo "mesh_leg rc=$?"; cat gpurun_out/r6b/mesh_leg.json | cut -c1-1500
