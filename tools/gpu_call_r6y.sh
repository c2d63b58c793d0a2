#!/bin/bash
# GPU call Y (round 6): weight-gradient launches three ways on one box, alternating - shipped (merged GEMM launches + one batched reduction),
# SDFHIP_WGRAD_SINGLE_LAUNCHES=1 (a launch per GEMM, batched reduction), SDFHIP_WREDUCE_PER_GEMM=1 (rounds 1 - 6) - after the tests of the new code.
O=gpurun_out/r6y
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
( timeout 1500 python -m pytest tests/test_gpu_bitrepro.py tests/test_gpu_northstar.py tests/test_gpu_bench_multirank.py -q -m gpu -k "batched or sensor or rgbd or bench" > $O/pytest_new.log 2>&1 ); echo "pytest new rc=$?"; tail -6 $O/pytest_new.log | cut -c1-400
B="python bench.py --no-cpu-baseline --no-config5 --no-bigmlp --no-preset --no-neus-acc --no-dense-sdf --no-mesh --no-volsdf --no-config4 --no-exchange-n1 --no-forward-only --steps 20 --warmup 5"
for V in per_gemm single shipped per_gemm2 single2 shipped2; do
  unset SDFHIP_WREDUCE_PER_GEMM SDFHIP_WGRAD_SINGLE_LAUNCHES
  case $V in per_gemm*) export SDFHIP_WREDUCE_PER_GEMM=1;; single*) export SDFHIP_WGRAD_SINGLE_LAUNCHES=1;; esac
  timeout 300 $B > $O/bench_$V.json 2> $O/bench_$V.err
  python - $O/bench_$V.json $V <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=json.load(open('gpurun_out/bench_detail.json'))['kernels']
    print(sys.argv[2], 'ms_per_step', d['ms_per_step'], 'wgrad', k.get('wgrad_kernel'), 'wreduce', k.get('wreduce_kernel'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
unset SDFHIP_WREDUCE_PER_GEMM SDFHIP_WGRAD_SINGLE_LAUNCHES
for V in per_gemm shipped; do
  unset SDFHIP_WREDUCE_PER_GEMM
  case $V in per_gemm*) export SDFHIP_WREDUCE_PER_GEMM=1;; esac
  timeout 300 python bench.py --config 5 --levels 16 --no-cpu-baseline --no-forward-only --steps 20 --warmup 5 > $O/bench5_$V.json 2> $O/bench5_$V.err
  python - $O/bench5_$V.json $V <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    k=json.load(open('gpurun_out/bench_detail.json'))['kernels']
    print('config5 l16', sys.argv[2], 'ms_per_step', d['ms_per_step'], {n:v for n,v in k.items() if 'wgrad' in n or 'wreduce' in n})
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
