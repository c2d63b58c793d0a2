#!/bin/bash
# round 5, GPU call L: which kernel runs under SDFHIP_PAIR_SDF=1 in the dense-SDF leg, and its launch time by rocprofv3 (kernel trace stats)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
for M in 0 1; do
  SDFHIP_PAIR_SDF=$M timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$M -o kt -- python $GRAFT_REPO_ROOT/bench.py --only inference --steps 3 > $O/kt$M.log 2>&1
  echo "-- PAIR=$M"; grep -h "geo_fwd_kernel\|geo_sdf_pair\|geo_encode" $O/kt$M/kt_kernel_stats.csv | cut -c1-200
  rm -f $O/kt$M/*kernel_trace.csv
done
