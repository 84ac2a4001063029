#!/bin/bash
# A/B of owner-tile variants on one box: tools/probes/owner_sweep.sh builds, this runs tools/kbench.py under rocprofv3 for each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in base "$@"; do
  if [ $tag = base ]; then unset MVSTER_LIB_PATH; else export MVSTER_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/build/libmvster_$tag.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag -- python3 tools/kbench.py --reps 6 > gpurun_out/ab_$tag.log 2>&1
  echo "== $tag"; python tools/kstats.py gpurun_out/ab_$tag 250 | grep "${AB_GREP:-coef_kernel<8\|owner_c8\|fwd_nhwc_kernel<8}"; tail -1 gpurun_out/ab_$tag.log
done
