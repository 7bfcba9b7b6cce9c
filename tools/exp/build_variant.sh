#!/bin/bash
# Build an experimental variant of libeco_hip.so: one source recompiled with extra -D flags, the rest linked from
# the normal build.  Usage: tools/exp/build_variant.sh <name> <source stem, e.g. eco_stem> <flags...>
#   -> tools/exp/libeco_hip_<name>.so (git-ignored; shipped to the GPU box, where an experiment script copies it over
#   the package's libeco_hip.so)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/eco-efficient-video-understanding_amd/csrc
NAME=$1; SRC=$2; shift 2
make -s -C $CSRC -j8 all
OBJS=""
for f in eco_api eco_conv eco_ops eco_wino eco_blocked eco_wgemm eco_stem eco_stemb; do
  if [ $f = $SRC ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -fno-slp-vectorize "$@" -c $CSRC/$f.hip -o /tmp/${f}_$NAME.o
    OBJS="$OBJS /tmp/${f}_$NAME.o"
  else
    OBJS="$OBJS $CSRC/build/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/exp/libeco_hip_$NAME.so
echo built tools/exp/libeco_hip_$NAME.so
