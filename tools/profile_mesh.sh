#!/bin/bash
# Evidence run for the mesh library (NOT yet run: round 5's GPU time ended with tools/gpu_call_r5p.sh): rocprofv3 kernel-trace stats of the
# standalone check (no Python start-up), then separate --pmc passes for HBM bytes and issue counters - no trace domains beside --pmc.
#   tools/profile_mesh.sh <tag>   ->  gpurun_out/<tag>/{kt,pmc_fetch,pmc_write,pmc_sq}; copy the *_kernel_stats.csv / counter csv you quote to profiles/
TAG=${1:-r6_mesh}
R=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/tests/_bin/mesh_gpu_check $GRAFT_REPO_ROOT/sdfstudio_amd/libsdfmesh.so $GRAFT_REPO_ROOT/tests/_bin/mesh_cases.bin"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o kt -- $B $R/check_kt.jsonl > $R/kt.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -o p -- $B $R/check_fetch.jsonl > $R/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -o p -- $B $R/check_write.jsonl > $R/pmc_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/pmc_sq -o p -- $B $R/check_sq.jsonl > $R/pmc_sq.log 2>&1
ls -R $R | head -40
