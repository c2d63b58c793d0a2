#!/bin/bash
# GPU call P (round 5, last 1.5 GPU-minutes): the standalone hardware check of libsdfmesh.so - no Python start-up, seconds of box time.
mkdir -p gpurun_out
rm -f gpurun_out/mesh_gpu_check.jsonl
timeout 9 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/mesh_gpu_check.jsonl
echo "mesh_gpu_check rc=$?"
cat gpurun_out/mesh_gpu_check.jsonl
