#!/bin/bash
# GPU call P (round 6): (1) mesh count call with the totals written through host-mapped memory - A/B against the copy in the same process image;
# (2) evidence for the mesh library as it is now (C-ABI check, rocprofv3 kernel stats + PMC passes, binding tests);
# (3) rocprofv3 kernel stats of the BASELINE config 1 / config 4 / NeuS-acc legs alone (bench.py --only ...).
mkdir -p gpurun_out/r6p
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  SDFMESH_NO_HOST_MAPPED=1 timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6p/check_copy_$i.jsonl > /dev/null 2>&1
  echo "copy   rc=$? $(grep -h '"crop512"' gpurun_out/r6p/check_copy_$i.jsonl | tail -1 | cut -c1-120)"
  timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6p/check_mapped_$i.jsonl > /dev/null 2>&1
  echo "mapped rc=$? $(grep -h '"crop512"' gpurun_out/r6p/check_mapped_$i.jsonl | tail -1 | cut -c1-120)"
done
cp gpurun_out/r6p/check_mapped_1.jsonl gpurun_out/r6p/mesh_gpu_check.jsonl
timeout 600 bash tools/profile_mesh.sh r6p/mesh > gpurun_out/r6p/profile_mesh.log 2>&1; echo "profile_mesh rc=$?"
timeout 1500 python -m pytest tests/test_gpu_zy_mesh_abi.py tests/test_gpu_zz_mesh.py -x -q -m gpu > gpurun_out/r6p/pytest_mesh.log 2>&1; echo "pytest mesh rc=$?"; tail -2 gpurun_out/r6p/pytest_mesh.log
for i in 1 2; do timeout 300 python tools/mesh_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mesh_leg', d['ms'], d['ms_wall_incl_host'], d['roofline']['frac'])"; done
R=$GRAFT_REPO_ROOT/gpurun_out/r6p
cd /tmp && export TMPDIR=/tmp
for leg in volsdf config4 neus_acc; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_$leg -o kt -- python $GRAFT_REPO_ROOT/bench.py --only $leg > $R/kt_$leg.log 2>&1
  echo "$leg rc=$?"; tail -1 $R/kt_$leg.log | cut -c1-400
  rm -f $R/kt_$leg/*kernel_trace.csv
done
