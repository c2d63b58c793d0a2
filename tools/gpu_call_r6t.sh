#!/bin/bash
# GPU call T (round 6): the eval-side caller (whole-image rays, get_outputs_for_camera_ray_bundle) - tests and the image figure of the inference leg.
mkdir -p gpurun_out/r6t
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_edge_cases.py -q -m gpu -x --no-header -p no:cacheprovider > gpurun_out/r6t/pytest.log 2>&1; echo "rc=$?"
grep -v Warning gpurun_out/r6t/pytest.log | grep -E "^E  |passed|failed|Error" | cut -c1-300 | head -20
timeout 600 python bench.py --only inference --steps 3 > gpurun_out/r6t/inference.json 2> gpurun_out/r6t/inference.err; echo "inference rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6t/inference.json').read().strip().splitlines()[-1])
fo=d['forward_only']; print('forward_only', fo['ms_per_batch'], fo['value'], fo.get('image_384x384'))
P
