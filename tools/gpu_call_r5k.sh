#!/bin/bash
# round 5, GPU call K: the pair-wave form of the sdf-only forward (csrc/pair_kernels.h, SDFHIP_PAIR_SDF=1) - its parity / reproducibility
# test and the same-box A/B of the dense-SDF leg (one library, the switch alternated)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
timeout 200 python -m pytest tests -m gpu -q -x -k "pair_wave or dense_grid" > $O/pytest_pair.log 2>&1
echo "pair test rc $?"; tail -4 $O/pytest_pair.log | cut -c1-400; grep "^E  " $O/pytest_pair.log | head -5 | cut -c1-300
for M in 0 1 0 1; do
  SDFHIP_PAIR_SDF=$M timeout 120 python bench.py --only inference --steps 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['dense_sdf']; f=d['forward_only']
print('PAIR=$M dense', s['ms'], s['value'], s['kernels_ms'], s['roofline']['frac'], 'fwd_only', f['ms_per_batch'])" | tee -a $O/ab_pair_sdf.txt
done
