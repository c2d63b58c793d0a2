#!/bin/bash
# GPU call L (round 6): the default bench run with the new legs + compact line; then the whole GPU suite.
mkdir -p gpurun_out/r6l
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python bench.py > gpurun_out/r6l/bench_line.json 2> gpurun_out/r6l/bench.err ) 2> gpurun_out/r6l/bench_time.txt
echo "bench rc=$?"; wc -c gpurun_out/r6l/bench_line.json; cat gpurun_out/r6l/bench_line.json; tail -3 gpurun_out/r6l/bench_time.txt
cp gpurun_out/bench_detail.json gpurun_out/r6l/bench_detail.json 2>/dev/null
grep -v "^\[bench\] GPU leg done" gpurun_out/r6l/bench.err | tail -15
( time timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r6l/pytest_gpu.log 2>&1 ) 2> gpurun_out/r6l/pytest_time.txt
echo "pytest rc=$?"; tail -15 gpurun_out/r6l/pytest_gpu.log; tail -3 gpurun_out/r6l/pytest_time.txt
