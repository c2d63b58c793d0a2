#!/bin/bash
# round 5, GPU call D (after the container was re-created: the outputs of calls B and C were lost with it): box class, the one-wave /
# pair-wave probes on this box, full GPU suite on the build with the no_grad fix, the default bench line (all legs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
{
  for p in probe_split probe_split_f16 probe_pair_sb1b4 probe_pair_abl1 probe_pair_abl2 probe_split_abl4 probe_split_relu; do
    echo "-- $p"; timeout 60 tools/_bin/$p
  done
} > $O/probe_pair_ab.txt 2>&1
grep -i "cyc" $O/probe_pair_ab.txt | head -20
timeout 1300 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep "^FAILED" $O/pytest.log | head -30
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"; tail -c 400 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5d/bench.json').read().strip().splitlines()[-1])
k=d['kernels']
print('train ms/step', d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>=0.1})
print('roofline', d['roofline']['frac'], 'cpu', d.get('cpu_baseline'))
f=d['forward_only']; print('forward_only', f['ms_per_batch'], f['roofline']['frac'], f['kernels_ms_per_batch'])
print('dense', d['dense_sdf']['ms'], d['dense_sdf']['value'], d['dense_sdf']['roofline']['frac'], d['dense_sdf']['kernels_ms'])
print('cfg5', {n:(v['ms_per_step']) for n,v in d['config5'].items() if isinstance(v,dict)})
print('bigmlp', {n:(v['ms_per_step'], v.get('ratio_to_256_wide_step'), v['kernels_ms_per_step']) for n,v in d['bigmlp'].items() if isinstance(v,dict)})
p=d['preset']; print('preset', p['ms_per_step'], p['iters_per_sec'], p['enqueue_vs_gpu'], p.get('native_kernel_ms_per_step'))
a=d['neus_acc']; print('neus_acc', a['ms_per_step'], a.get('samples_kept_per_ray'), a['enqueue_vs_gpu'])
PY
