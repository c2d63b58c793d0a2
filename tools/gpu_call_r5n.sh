#!/bin/bash
# round 5, GPU call N: the default bench line of the tree as committed (bench.py gained fields after call H; same library)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5n/bench.json').read().strip().splitlines()[-1])
print('train ms/step', d['ms_per_step'], 'digest', d['library_digest'], 'stale', d['roofline']['traffic_stale'], d['step_roofline']['hbm_pmc_stale'])
s=d['dense_sdf']['roofline']; print('dense', d['dense_sdf']['ms'], s['frac'], s['frac_whole_leg'], s['frac_kernel_at_full_network_G'], s['traffic_stale'])
f=d['forward_only']; print('fo', f['ms_per_batch'], f['roofline']['frac'], f['roofline']['traffic_stale'])
print('cfg5', {n:(v['ms_per_step'], v['roofline'].get('traffic_stale')) for n,v in d['config5'].items() if isinstance(v,dict)})
print('acc', d['neus_acc']['ms_per_step'], 'preset', d['preset']['ms_per_step'], 'bigmlp', d['bigmlp']['config2_batch']['ratio_to_256_wide_step'])
PY
