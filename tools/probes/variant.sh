#!/bin/bash
# Build a variant of libmvster_hip.so with extra -D macros in chosen translation units (for MVSTER_LIB_PATH A/B runs):
#   tools/probes/variant.sh TAG "-DFOO=1 -DBAR" conv_mfma [more units]   -> tools/probes/build/libmvster_TAG.so
set -e
cd "$(dirname "$0")/../.."
CSRC=deep_reconstruction_with_epipolar_lines_mvster_amd/csrc
OUT=tools/probes/build
mkdir -p $OUT
TAG=$1; DEFS=$2; shift 2
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I include -I $CSRC"
objs=""; skip=""
for f in "$@"; do
  hipcc $FLAGS $DEFS -c $CSRC/$f.hip -o $OUT/${f}_$TAG.o &
  objs="$objs $OUT/${f}_$TAG.o"; skip="$skip\|/$f.o"
done
wait
others=$(ls $CSRC/build/*.o | grep -v "NOTHING$skip")
hipcc -shared --offload-arch=gfx950 -o $OUT/libmvster_$TAG.so $objs $others
rm -f $objs
echo built $OUT/libmvster_$TAG.so
