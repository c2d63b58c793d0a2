#!/bin/bash
# Runs ON the GPU box: classify it; on a slow-class box measure the current build against the per-layer
# unrolled build of round 2's start (tools/_bin/libsdfhip_unroll.so, if present) and keep the full bench line.
TAG=${1:-fish}
R=gpurun_out/$TAG
mkdir -p $R
bash tools/box_class.sh $R/box.log | tail -1
if grep -q "BOX CLASS: SLOW" $R/box.log; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/bench_slowbox.json 2> $R/bench.err
  LIBS="sdfstudio_amd/libsdfhip.so"
  [ -f tools/_bin/libsdfhip_unroll.so ] && LIBS="$LIBS tools/_bin/libsdfhip_unroll.so"
  bash tools/ab.sh 10 $LIBS 2>&1 | tee $R/ab.log | tail -4
  timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -2 | tee $R/gpu_tests.log
fi
