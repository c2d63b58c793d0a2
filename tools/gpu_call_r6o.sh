#!/bin/bash
# GPU call O (round 6): mesh library with the two emit passes in one launch - C-ABI check, rocprofv3 + PMC passes, binding tests, the bench leg.
mkdir -p gpurun_out/r6o
timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6o/mesh_gpu_check.jsonl
echo "mesh_gpu_check rc=$?"; tail -3 gpurun_out/r6o/mesh_gpu_check.jsonl
timeout 600 bash tools/profile_mesh.sh r6o/mesh > gpurun_out/r6o/profile_mesh.log 2>&1; echo "profile rc=$?"
timeout 1500 python -m pytest tests/test_gpu_zy_mesh_abi.py tests/test_gpu_zz_mesh.py -x -q -m gpu > gpurun_out/r6o/pytest_mesh.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6o/pytest_mesh.log
for i in 1 2 3; do timeout 300 python tools/mesh_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms'], d['ms_wall_incl_host'], d['roofline']['frac'])"; done
