#!/bin/bash
# Build an experimental variant of libeco_hip.so: one source recompiled with extra -D flags, the rest linked from
# the normal build.  Usage: tools/exp/build_variant.sh <name> <source stem, e.g. eco_stem> <flags...>
#   -> tools/exp/libeco_hip_<name>.so (git-ignored; shipped to the GPU box, where an experiment script copies it over
#   the package's libeco_hip.so)
# The probe instrumentation of rounds 3-4 (-DECO_SPANP_PROBE=bits, -DECO_SPANP_TS, -DECO_STEMB_PROBE=..., -DECO_WFUSED_PROBE=...,
# -DECO_WGEMM_PROBE=..., -DECO_STEM_PROBE=..., -DECO_EPI_PROBE_NOSTORE) is NOT in the product sources any more (round 5:
# tools/strip_probes.py took it out, the compiled device code is identical): it lives in tools/exp/probes.patch, which is
# applied here to a scratch copy of csrc/ before the variant is compiled.  The patch was cut against the round-5 sources;
# if a kernel has moved since, re-base the hunk it complains about.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/eco-efficient-video-understanding_amd/csrc
NAME=$1; SRC=$2; shift 2
make -s -C $CSRC -j8 all
SCRATCH=$(mktemp -d /tmp/eco_probe_XXXX)
mkdir -p $SCRATCH/eco/csrc $SCRATCH/include
cp $CSRC/*.hip $CSRC/*.h $SCRATCH/eco/csrc/
cp $ROOT/include/eco_hip.h $SCRATCH/include/
(cd $SCRATCH/eco/csrc && patch -p1 -s < $ROOT/tools/exp/probes.patch)
OBJS=""
for f in eco_api eco_conv eco_ops eco_wino eco_blocked eco_blocked_ops eco_wgemm eco_wino3 eco_wino_s2 eco_stem eco_stemb; do
  if [ $f = $SRC ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=default -fno-slp-vectorize "$@" -c $SCRATCH/eco/csrc/$f.hip -o /tmp/${f}_$NAME.o
    OBJS="$OBJS /tmp/${f}_$NAME.o"
  else
    OBJS="$OBJS $CSRC/build/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/exp/libeco_hip_$NAME.so
rm -rf $SCRATCH
echo built tools/exp/libeco_hip_$NAME.so
