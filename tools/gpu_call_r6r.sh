#!/bin/bash
# GPU call R (round 6): the edge-case tests
mkdir -p gpurun_out/r6r
timeout 900 python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -x --no-header -p no:cacheprovider > gpurun_out/r6r/pytest_edge.log 2>&1; echo "rc=$?"
tail -60 gpurun_out/r6r/pytest_edge.log
