#!/bin/bash
# round 5, GPU call G: the evidence set of the final build - box class, rocprofv3 stats + PMC passes (training step, inference legs, config 5 at
# 16 and 8 levels), then the default bench line (its traffic fields read the summaries just written: same library digest), then the full GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O/profiles
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
head -12 $O/box_class.txt | tail -4
bash tools/profile_round.sh r5 > $O/prof_train.log 2>&1; tail -3 $O/prof_train.log | cut -c1-300
bash tools/profile_eval.sh r5_eval > $O/prof_eval.log 2>&1; tail -3 $O/prof_eval.log | cut -c1-300
bash tools/profile_round.sh r5_cfg5l16 --config 5 --levels 16 > $O/prof_cfg5l16.log 2>&1; tail -2 $O/prof_cfg5l16.log | cut -c1-300
bash tools/profile_round.sh r5_cfg5 --config 5 > $O/prof_cfg5.log 2>&1; tail -2 $O/prof_cfg5.log | cut -c1-300
cp profiles/r5_* $O/profiles/ 2>/dev/null
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5g/bench.json').read().strip().splitlines()[-1])
k=d['kernels']
print('train ms/step', d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>=0.1})
r=d['roofline']; print('roofline', r['frac'], r['traffic'], r['traffic_stale'], 'enc', d['encode_roofline']['frac'], d['encode_roofline'].get('traffic'), d['encode_roofline'].get('traffic_stale'), 'step', d['step_roofline'])
f=d['forward_only']; print('forward_only', f['ms_per_batch'], f['roofline']['frac'], f['roofline'].get('traffic'), f['roofline'].get('traffic_stale'), f['kernels_ms_per_batch'])
s=d['dense_sdf']; print('dense', s['ms'], s['value'], s['roofline']['frac'], s['roofline'].get('traffic'), s['roofline'].get('traffic_stale'), s['kernels_ms'])
print('cfg5', {n:(v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('traffic'), v['roofline'].get('traffic_stale')) for n,v in d['config5'].items() if isinstance(v,dict)})
print('bigmlp', {n:(v['ms_per_step'], v.get('ratio_to_256_wide_step')) for n,v in d['bigmlp'].items() if isinstance(v,dict)})
p=d['preset']; print('preset', p['ms_per_step'], p['iters_per_sec'], p['enqueue_vs_gpu'])
a=d['neus_acc']; print('neus_acc', a['ms_per_step'], a.get('samples_kept_per_ray'), a['enqueue_vs_gpu'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep "^FAILED" $O/pytest.log | head -30
