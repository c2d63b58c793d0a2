#!/bin/bash
# Builds the micro-benchmarks under tools/ into tools/_bin/ (git-ignored; they travel to the GPU box with the gpurun snapshot).
# tools/box_class.sh uses them to classify a box and to run the diagnostic battery on slow-class boxes.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/_bin
mkdir -p $OUT
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -I $ROOT/sdfstudio_amd/csrc"
b() { echo "building $1"; $CC ${@:3} -o $OUT/$1 $ROOT/tools/$2 2>&1 | grep -E "error" || true; }
b probe_split probe_split.hip &
b probe_split_looped probe_split.hip -DPROBE_LOOPED &
b probe_split_f16 probe_split.hip -DPROBE_F16 &
b probe_split_f16_pkrtz probe_split.hip -DPROBE_F16 -DPROBE_PKRTZ &
wait
b probe_split_relu probe_split.hip -DPROBE_RELU &
b probe_split_abl4 probe_split.hip -DABL=4 &
b probe_icache probe_icache.hip &
b probe_box probe_box.hip &
wait
b probe_pair_sb1b4 probe_pair.hip -DPAIR_SB=1 -DPAIR_BATCH=4 &
b probe_pair_abl1 probe_pair.hip -DPAIR_SB=1 -DPAIR_BATCH=4 -DPAIR_ABL=1 &
b probe_pair_abl2 probe_pair.hip -DPAIR_SB=1 -DPAIR_BATCH=4 -DPAIR_ABL=2 &
b probe_atomic probe_atomic.hip &
wait
ls -la $OUT | grep probe
