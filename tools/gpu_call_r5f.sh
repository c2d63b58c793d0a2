#!/bin/bash
# round 5, GPU call F (short): every comparison of config 5's steady-state bars with the compensated 24-bit sdf rows (fp32-class ratios included)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
for HP in 1 0; do
SDFHIP_NUMFIELD_HP=$HP SDFHIP_TEST_KEEP_GOING=1 SDFHIP_TEST_LOG=$PWD/$O/cfg5_bars_hp$HP.log timeout 300 python -m pytest tests/test_gpu_config5.py -q -k "mask16 or step200000" > $O/cfg5_keepgoing_hp$HP.log 2>&1
echo "-- HP=$HP"; cut -c1-330 $O/cfg5_bars_hp$HP.log | sed 's/tests.test_gpu_config5.py:://'
done
