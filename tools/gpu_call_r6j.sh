#!/bin/bash
# GPU call J (round 6): ref-nerf tests (flip-aware), hooks, bounded NeuS-acc, existing NeuS-acc parity, then the bench's neus_acc leg.
mkdir -p gpurun_out/r6j
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests/test_gpu_refnerf.py tests/test_gpu_field_hooks.py tests/test_gpu_neus_acc_bounded.py "tests/test_gpu_parity.py::test_neus_acc_model_packed_path" -q -m gpu > gpurun_out/r6j/pytest.log 2>&1
echo "pytest rc=$?"; grep -v Warning gpurun_out/r6j/pytest.log | tail -25
timeout 900 python bench.py --no-config5 --no-bigmlp --no-preset --no-dense-sdf --no-mesh --no-volsdf --no-config4 --no-exchange-n1 --no-cpu-baseline --no-forward-only > gpurun_out/r6j/bench_acc.json 2> gpurun_out/r6j/bench_acc.err
echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/bench_detail.json')); print(json.dumps(d['neus_acc'])[:2500]); print(d['ms_per_step'], d['enqueue_vs_gpu'])"
tail -3 gpurun_out/r6j/bench_acc.err
