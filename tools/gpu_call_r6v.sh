#!/bin/bash
# GPU call V (round 6): the weight stream on the BUFFER form of the LDS-DMA (exact s_waitcnt counts come back): bench first (the A/B against call U's
# line on the old form), then bit-reproducibility + north-star + parity tests, then the rest of the suite.
O=gpurun_out/r6v
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
( time timeout 900 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$?"; python - <<'P'
import json
d=json.load(open('gpurun_out/r6v/bench_line.json'))
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernels', d['kernels_ms_per_step'])
L=d['legs']
for k in ('config5_levels8','config5_levels16','bigmlp_config2_batch','preset','neus_acc','volsdf_rays4096','config4','forward_only','dense_sdf'):
    v=L.get(k) or {}
    print(k, v.get('ms', v.get('ms_per_step')), v.get('frac'))
P
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
( time timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest_time.txt
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log; tail -3 $O/pytest_time.txt
