#!/bin/bash
# GPU call F (round 6): mesh library after the emit restructuring (4 lanes per vertex, one thread per face index): check + profile.
mkdir -p gpurun_out/r6f
timeout 60 tests/_bin/mesh_gpu_check sdfstudio_amd/libsdfmesh.so tests/_bin/mesh_cases.bin gpurun_out/r6f/mesh_gpu_check.jsonl
echo "mesh_gpu_check rc=$?"
tail -4 gpurun_out/r6f/mesh_gpu_check.jsonl
timeout 600 bash tools/profile_mesh.sh r6f/mesh > gpurun_out/r6f/profile_mesh.log 2>&1
echo "profile_mesh rc=$?"
