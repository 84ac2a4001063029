#!/bin/bash
# Build variants of libmvster_hip.so that differ in the C = 8 owner kernel's tile width and samples per lane, for
# tools/kbench.py to time (MVSTER_LIB_PATH selects the build).  usage: tools/probes/owner_sweep.sh  (in the build container)
set -e
cd "$(dirname "$0")/../.."
CSRC=deep_reconstruction_with_epipolar_lines_mvster_amd/csrc
OUT=tools/probes/build
mkdir -p $OUT
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I include -I $CSRC"
# DEFS="-DMVSTER_C16_KPT=4 -DMVSTER_STAGED_KEEP_C=8" TAG=c16_kpt4 tools/probes/owner_sweep.sh builds one variant with those macros
if [ -n "$DEFS" ]; then
  objs=""
  for f in cost_volume_fwd cost_volume_bwd cost_volume_binned; do
    hipcc $FLAGS $DEFS -c $CSRC/$f.hip -o $OUT/${f}_$TAG.o &
    objs="$objs $OUT/${f}_$TAG.o"
  done
  wait
  others=$(ls $CSRC/build/*.o | grep -v "cost_volume_fwd.o\|cost_volume_bwd.o\|cost_volume_binned.o")
  hipcc -shared --offload-arch=gfx950 -o $OUT/libmvster_$TAG.so $objs $others
  echo built $OUT/libmvster_$TAG.so
  exit 0
fi
for v in ${SWEEP:-"3 4" "4 3" "4 6" "5 2" "5 3"}; do
  set -- ${v//_/ }
  tag=lw$1_kpt$2${3:+_okpt$3}
  objs=""
  for f in cost_volume_fwd cost_volume_bwd cost_volume_binned; do
    hipcc $FLAGS -DMVSTER_C8_LW=$1 -DMVSTER_C8_KPT=$2 ${3:+-DMVSTER_OWNER_KPT=$3} -c $CSRC/$f.hip -o $OUT/${f}_$tag.o &
    objs="$objs $OUT/${f}_$tag.o"
  done
  wait
  others=$(ls $CSRC/build/*.o | grep -v "cost_volume_fwd.o\|cost_volume_bwd.o\|cost_volume_binned.o")
  hipcc -shared --offload-arch=gfx950 -o $OUT/libmvster_$tag.so $objs $others
  echo built $OUT/libmvster_$tag.so
done
