#!/bin/bash
# round 5, GPU call E: the first-order backward fix (accIn kept alive through the skip layer's in0 gemm) on the tests it broke, the
# compensated 24-bit sdf row on config 5's steady-state bars (raw numbers of every comparison), its cost, and the one-launch eval forward A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -k "forward_geonetwork_is_differentiable or numerical_gradient_field_against or background_mlp_models or nerf_background_field or no_grad_forward or northstar_bars_config5 or config5_numerical_gradient_step_is_bit or background_grid or angelo" > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep "^FAILED" $O/pytest.log | head -30
SDFHIP_TEST_KEEP_GOING=1 SDFHIP_TEST_LOG=$PWD/$O/cfg5_bars.log timeout 300 python -m pytest tests/test_gpu_config5.py -q -k "northstar_bars and mask16" > $O/cfg5_keepgoing.log 2>&1
cat $O/cfg5_bars.log | cut -c1-420
SDFHIP_NUMFIELD_HP=0 SDFHIP_TEST_KEEP_GOING=1 SDFHIP_TEST_LOG=$PWD/$O/cfg5_bars_hp0.log timeout 300 python -m pytest tests/test_gpu_config5.py -q -k "northstar_bars and mask16" > $O/cfg5_keepgoing_hp0.log 2>&1
echo "-- HP=0 (22-bit evaluations)"; cat $O/cfg5_bars_hp0.log | cut -c1-300
for rep in 1 2; do for L in sdfstudio_amd/libsdfhip.so tools/_bin/libsdfhip_single.so; do
  SDFHIP_LIB=$PWD/$L timeout 300 python bench.py --only inference --steps 3 2>/dev/null | tail -1 >> $O/ab_single_launch_eval.jsonl; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r5e/ab_single_launch_eval.jsonl'):
    d=json.loads(l); f=d['forward_only']; print(d['library'].split('/')[-1], 'fwd_only', f['ms_per_batch'], f['kernels_ms_per_batch'].get('geo_fwd_kernel'), f['kernels_ms_per_batch'].get('col_fwd_kernel'), f['roofline']['frac'], 'dense', d['dense_sdf']['ms'])
PY
for HP in 1 0; do
  SDFHIP_NUMFIELD_HP=$HP timeout 300 python bench.py --config 5 --levels 16 --no-cpu-baseline --no-forward-only 2>/dev/null | tail -1 > $O/cfg5l16_hp$HP.json
  python - $HP <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r5e/cfg5l16_hp{sys.argv[1]}.json').read())
k=d['kernels']; print('cfg5 l16 HP', sys.argv[1], d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>=0.05})
PY
done
