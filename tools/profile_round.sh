#!/bin/bash
# Evidence run on the GPU box (via gpurun): kernel-trace stats of the default bench command, then the PMC passes the guide
# prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE passes; no sys/hip/hsa trace domains next to --pmc),
# then SQ busy / wait counters and the LDS bank-conflict counters.   tools/profile_round.sh <tag> [bench.py args]   ->  gpurun_out/<tag>/{kt,pmc_fetch,pmc_write,pmc_sq}
TAG=${1:-r1}
shift
EXTRA="$@"   # extra bench.py arguments, e.g. --config 5 (tag it r3_cfg5: bench.py picks the cfg5 summaries for its config-5 line)
R=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-bigmlp --no-preset --no-neus-acc --no-dense-sdf --no-mesh --no-volsdf --no-config4 --no-exchange-n1 $EXTRA"  # the headline leg alone (the appended config-5 / 512-wide legs would pollute the per-step sums)
BP="$B --no-forward-only --no-kernel-table"  # PMC passes: training steps only, so launches / steps = launches per step
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o kt -- $B --steps 10 --warmup 3 > $R/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -o p -- $BP --steps 2 --warmup 1 > $R/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -o p -- $BP --steps 2 --warmup 1 > $R/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $R/pmc_sq -o p -- $BP --steps 2 --warmup 1 > $R/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/pmc_lds -o p -- $BP --steps 2 --warmup 1 > $R/pmc_lds.log 2>&1
rm -f $R/*/*kernel_trace.csv   # large; the stats and counter files carry what the summaries need
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py $TAG
