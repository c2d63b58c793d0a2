#!/bin/bash
# round 5, GPU call A: new producer (Softplus form, fma_mix split) vs the round-4 producer - parity subset, inference A/B, PMC of the inference kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
bash tools/box_class.sh $O/box_class.txt > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "field_small or model_against_reference_golden or field_forward_is_bit_reproducible or northstar_bars or dense_grid or field_full_size or test_full_size_properties or fused_training_step_is_bit or background_mlp_models" > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
for rep in 1 2; do for L in sdfstudio_amd/libsdfhip.so tools/_bin/libsdfhip_old.so; do
  SDFHIP_LIB=$PWD/$L timeout 300 python bench.py --only inference --steps 3 2>/dev/null | tail -1 >> $O/ab_infer.jsonl; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r5a/ab_infer.jsonl'):
    d=json.loads(l); print(d['library'].split('/')[-1], 'dense', d['dense_sdf']['ms'], d['dense_sdf']['kernels_ms'], d['dense_sdf']['roofline']['frac'], 'fwd', d['forward_only']['ms_per_batch'], d['forward_only']['kernels_ms_per_batch'].get('geo_fwd_kernel'), d['forward_only']['kernels_ms_per_batch'].get('col_fwd_kernel'), d['forward_only']['roofline']['frac'])
PY
EVAL_SQ_ONLY=1 bash tools/profile_eval.sh r5a_eval_old $PWD/tools/_bin/libsdfhip_old.so > $O/prof_old.log 2>&1
bash tools/profile_eval.sh r5a_eval > $O/prof_new.log 2>&1
tail -8 $O/prof_new.log
for L in sdfstudio_amd/libsdfhip.so tools/_bin/libsdfhip_old.so; do
  SDFHIP_LIB=$PWD/$L timeout 400 python bench.py --no-cpu-baseline --no-config5 --no-bigmlp --no-dense-sdf 2>/dev/null | tail -1 >> $O/ab_train.jsonl; done
python - <<'PY'
import json
for l in open('gpurun_out/r5a/ab_train.jsonl'):
    d=json.loads(l); k=d['kernels']; print('train ms/step', d['ms_per_step'], {n.replace('_kernel',''):round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>=0.1})
PY
