#!/bin/bash
# round 5, GPU call B: full GPU suite on the build with the producer trim, zero-scratch kernels, rolled prop_bwd, 24-bit numerical-normal
# evaluations; then the default bench line (all legs), and the config-5 steady-state step with the 24-bit evaluations on / off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc $?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5b/bench.json').read().strip().splitlines()[-1])
k=d['kernels']
print('train ms/step', d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>=0.1})
print('forward_only', d['forward_only']['ms_per_batch'], d['forward_only']['roofline']['frac'])
print('dense', d['dense_sdf']['ms'], d['dense_sdf']['value'], d['dense_sdf']['roofline']['frac'])
print('cfg5', {n:(v['ms_per_step']) for n,v in d['config5'].items() if isinstance(v,dict)})
print('bigmlp', {n:(v['ms_per_step'], v.get('ratio_to_256_wide_step')) for n,v in d['bigmlp'].items() if isinstance(v,dict)})
p=d['preset']; print('preset', p['ms_per_step'], p['iters_per_sec'], p['enqueue_vs_gpu'], p['native_kernel_ms_per_step'])
a=d['neus_acc']; print('neus_acc', a['ms_per_step'], a['samples_kept_per_ray'], a['enqueue_vs_gpu'])
PY
for HP in 1 0; do
  SDFHIP_NUMFIELD_HP=$HP timeout 300 python bench.py --config 5 --levels 16 --no-cpu-baseline --no-forward-only 2>/dev/null | tail -1 > $O/cfg5l16_hp$HP.json
  python - $HP <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r5b/cfg5l16_hp{sys.argv[1]}.json').read())
k=d['kernels']; print('cfg5 l16 HP', sys.argv[1], d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>=0.05})
PY
done
