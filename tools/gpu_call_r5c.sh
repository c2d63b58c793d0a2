#!/bin/bash
# round 5, GPU call C: full GPU suite (no -x) on the build with half-width 512-wide kernels; the config-5 bars' raw numbers; bigmlp ratio
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep "^FAILED" $O/pytest.log | head -30
SDFHIP_TEST_KEEP_GOING=1 SDFHIP_TEST_LOG=$PWD/$O/cfg5_bars.log timeout 300 python -m pytest tests/test_gpu_config5.py -q -k "northstar_bars and mask16" > $O/cfg5_keepgoing.log 2>&1
grep -h "fp64\|rows where" $O/cfg5_keepgoing.log | head -30
timeout 400 python bench.py --no-cpu-baseline --no-config5 --no-preset --no-neus-acc --no-dense-sdf --no-forward-only > $O/bench_bigmlp.json 2> $O/bench_bigmlp.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench_bigmlp.json').read().strip().splitlines()[-1])
print('train ms/step', d['ms_per_step'])
for n,v in d['bigmlp'].items():
    if isinstance(v,dict): print('bigmlp', n, v['ms_per_step'], v.get('ratio_to_256_wide_step'), v['kernels_ms_per_step'])
PY
