#!/bin/bash
# GPU call S (round 6): the whole GPU suite after the empty-input changes (_lib.ptr / rawptr, theta_bar zero fill).
mkdir -p gpurun_out/r6s
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r6s/pytest_gpu.log 2>&1 ) 2> gpurun_out/r6s/pytest_time.txt
echo "pytest rc=$?"; tail -8 gpurun_out/r6s/pytest_gpu.log; tail -3 gpurun_out/r6s/pytest_time.txt
