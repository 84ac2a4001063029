#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...> -- <python script + args>   (run on the GPU box via gpurun)
# One rocprofv3 --pmc pass (kernel-trace only, never combined with sys/hip traces) into gpurun_out/pmc_<tag>/.
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag exit=$?"
