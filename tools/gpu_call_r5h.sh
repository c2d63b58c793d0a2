#!/bin/bash
# round 5, GPU call H: the default bench line once more with the evidence files of call G in place (traffic fields pick the digest-matching
# summaries), and the 500-step sustained run with clocks / power
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h/bench.json').read().strip().splitlines()[-1])
print('train ms/step', d['ms_per_step'], 'digest', d['library_digest'])
r=d['roofline']; print('roofline', r['frac'], r['traffic'], r['traffic_stale'], 'enc', d['encode_roofline']['frac'], d['encode_roofline'].get('traffic_stale'), 'step', d['step_roofline']['hbm_GB_per_step_pmc'], d['step_roofline']['hbm_pmc_stale'])
f=d['forward_only']; print('forward_only', f['ms_per_batch'], f['roofline']['frac'], f['roofline'].get('traffic'), f['roofline'].get('traffic_stale'))
s=d['dense_sdf']; print('dense', s['ms'], s['value'], s['roofline']['frac'], s['roofline'].get('traffic'), s['roofline'].get('traffic_stale'))
print('cfg5', {n:(v['ms_per_step'], v['roofline']['frac'], v['roofline'].get('traffic_stale')) for n,v in d['config5'].items() if isinstance(v,dict)})
PY
timeout 200 python tools/sustained_run.py $O/sustained.json --steps 500 > $O/sustained.log 2>&1; tail -3 $O/sustained.log | cut -c1-400
