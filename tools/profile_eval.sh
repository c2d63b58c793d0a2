#!/bin/bash
# Evidence run for the INFERENCE paths (dense SDF evaluation + eval-mode render: `bench.py --only inference`), same discipline as
# tools/profile_round.sh: kernel-trace stats, then separate --pmc passes (no trace domains beside --pmc).
#   tools/profile_eval.sh <tag> [lib.so]   ->  gpurun_out/<tag>/{kt,pmc_fetch,pmc_write,pmc_sq,pmc_lds}  ->  profiles/<tag>_{kernel_stats,pmc_summary}.csv
TAG=${1:-r5_eval}
LIB=$2
R=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export SDFHIP_LIB=$LIB
B="python $GRAFT_REPO_ROOT/bench.py --only inference"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o kt -- $B --steps 3 > $R/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/pmc_sq -o p -- $B --steps 1 > $R/pmc_sq.log 2>&1
if [ -z "$EVAL_SQ_ONLY" ]; then
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -o p -- $B --steps 1 > $R/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -o p -- $B --steps 1 > $R/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc_lds -o p -- $B --steps 1 > $R/pmc_lds.log 2>&1
fi
rm -f $R/*/*kernel_trace.csv
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py $TAG --inference
