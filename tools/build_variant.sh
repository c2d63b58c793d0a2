#!/bin/bash
# Build a libsdfhip variant with extra compiler flags for same-box A/B runs: tools/build_variant.sh <name> [-DFOO=1 ...]
# -> tools/_bin/libsdfhip_<name>.so
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/_bin; OBJ=$OUT/obj_$NAME; mkdir -p $OBJ
for f in api inst_a inst_a_fwd inst_a_inf inst_a_bwd inst_b inst_c inst_c_fwd inst_c_inf inst_c_bwd; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-result -Wno-unused-function "$@" -I $ROOT/sdfstudio_amd/csrc \
    -c $ROOT/sdfstudio_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/api.o $OBJ/inst_a.o $OBJ/inst_a_fwd.o $OBJ/inst_a_inf.o $OBJ/inst_a_bwd.o $OBJ/inst_b.o $OBJ/inst_c.o $OBJ/inst_c_fwd.o $OBJ/inst_c_inf.o $OBJ/inst_c_bwd.o -o $OUT/libsdfhip_$NAME.so
rm -rf $OBJ
echo built $OUT/libsdfhip_$NAME.so
