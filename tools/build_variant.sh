#!/bin/bash
# Build a libsdfhip variant with extra compiler flags for same-box A/B runs: tools/build_variant.sh <name> [-DFOO=1 ...]
# -> tools/_bin/libsdfhip_<name>.so   (sources and base flags come from sdfstudio_amd/build.py)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/_bin; OBJ=$OUT/obj_$NAME; mkdir -p $OBJ
SRCS=$(cd $ROOT && python -c "from sdfstudio_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))")
FLAGS=$(cd $ROOT && python -c "from sdfstudio_amd.build import FLAGS; print(' '.join(FLAGS))")
OBJS=""
for f in $SRCS; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -I $ROOT/sdfstudio_amd/csrc -c $ROOT/sdfstudio_amd/csrc/$f.hip -o $OBJ/$f.o &
  OBJS="$OBJS $OBJ/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libsdfhip_$NAME.so
rm -rf $OBJ
echo built $OUT/libsdfhip_$NAME.so
