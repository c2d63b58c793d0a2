#!/bin/bash
# A/B builds of libsdfhip on the SAME GPU box (timings are not comparable across boxes): tools/ab.sh <steps> <libA> <libB> ...
STEPS=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    SDFHIP_LIB=$L SDFHIP_BENCH_ALLOW_NONFINITE=${ALLOW_NONFINITE:-0} python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('$L'.split('/')[-1], 'ms/step', d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>=0.04})"
  done
done
