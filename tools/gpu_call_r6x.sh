#!/bin/bash
# GPU call X (round 6): the split-K reductions of a backward call in ONE launch (wreduce_batch_kernel) against the per-GEMM form
# (SDFHIP_WREDUCE_PER_GEMM=1), same box, alternating; the new tests (bit-identity of the two forms, sensor-depth losses, replicas check).
O=gpurun_out/r6x
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
( timeout 1500 python -m pytest tests/test_gpu_bitrepro.py tests/test_gpu_northstar.py tests/test_gpu_bench_multirank.py -x -q -m gpu -k "batched or sensor or rgbd or bench" > $O/pytest_new.log 2>&1 ); echo "pytest new rc=$?"; tail -4 $O/pytest_new.log
B="python bench.py --no-cpu-baseline --no-config5 --no-bigmlp --no-preset --no-neus-acc --no-dense-sdf --no-mesh --no-volsdf --no-config4 --no-exchange-n1 --no-forward-only --steps 20 --warmup 5"
for V in per_gemm batched per_gemm2 batched2; do
  case $V in per_gemm*) export SDFHIP_WREDUCE_PER_GEMM=1;; *) unset SDFHIP_WREDUCE_PER_GEMM;; esac
  timeout 300 $B > $O/bench_$V.json 2> $O/bench_$V.err
  python - $O/bench_$V.json $V <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d['kernels_ms_per_step']
print(sys.argv[2], 'ms_per_step', d['ms_per_step'], 'wgrad', k.get('wgrad_kernel'), 'wreduce', k.get('wreduce_kernel'), 'geo_bwd', k.get('geo_bwd_kernel'))
PY
done
unset SDFHIP_WREDUCE_PER_GEMM
# config 5 (numerical-gradient field: its own backward entry point) both ways
for V in per_gemm batched; do
  case $V in per_gemm*) export SDFHIP_WREDUCE_PER_GEMM=1;; *) unset SDFHIP_WREDUCE_PER_GEMM;; esac
  timeout 300 python bench.py --config 5 --levels 16 --no-cpu-baseline --no-forward-only --steps 20 --warmup 5 > $O/bench5_$V.json 2> $O/bench5_$V.err
  python - $O/bench5_$V.json $V <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d['kernels_ms_per_step']
print('config5 l16', sys.argv[2], 'ms_per_step', d['ms_per_step'], {n:v for n,v in k.items() if 'wgrad' in n or 'wreduce' in n})
PY
done
