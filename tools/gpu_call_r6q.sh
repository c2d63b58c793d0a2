#!/bin/bash
# GPU call Q (round 6): bench.py after the restructuring (appended legs run AFTER the line at N > 1): the default N = 1 run, the FULL configuration at
# N = 2 over gloo on the one GPU (control flow of the N > 1 order: line first, legs after), the multirank tests.
mkdir -p gpurun_out/r6q
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python bench.py > gpurun_out/r6q/bench_line.json 2> gpurun_out/r6q/bench.err ) 2> gpurun_out/r6q/bench_time.txt
echo "bench rc=$?"; wc -c gpurun_out/r6q/bench_line.json; cut -c1-300 gpurun_out/r6q/bench_line.json; tail -3 gpurun_out/r6q/bench_time.txt
cp gpurun_out/bench_detail.json gpurun_out/r6q/bench_detail.json 2>/dev/null
( time SDFHIP_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 5 --warmup 2 --no-bigmlp > gpurun_out/r6q/bench2_line.json 2> gpurun_out/r6q/bench2.err ) 2> gpurun_out/r6q/bench2_time.txt
echo "bench N=2 (gloo) rc=$?"; wc -l gpurun_out/r6q/bench2_line.json; cut -c1-300 gpurun_out/r6q/bench2_line.json; tail -3 gpurun_out/r6q/bench2_time.txt
grep "appended legs" gpurun_out/r6q/bench2.err | cut -c1-1500
cp gpurun_out/bench_detail.json gpurun_out/r6q/bench2_detail.json 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu > gpurun_out/r6q/pytest_multirank.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r6q/pytest_multirank.log
