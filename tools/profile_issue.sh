#!/bin/bash
# One more PMC pass for the fused kernels: where the issue slots go (LDS / VMEM / VALU / scalar active cycles, LDS bank
# conflicts, waits on LDS).  tools/profile_issue.sh <tag> -> gpurun_out/<tag>/pmc_issue ; reduce with tools/prof_issue_summary.py
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-table --steps 2 --warmup 1"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d $R/pmc_issue -o p -- $B > $R/pmc_issue.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC \
  --output-format csv -d $R/pmc_issue2 -o p -- $B > $R/pmc_issue2.log 2>&1
rm -f $R/*/*kernel_trace.csv
