#!/bin/bash
# Round-3 profile collection on the GPU box (gpurun): kernel stats, FETCH / WRITE / ATOMIC / SQ counter passes (separate runs,
# kernel-trace only), the un-profiled bench lines.  Copy the summaries you want judged from gpurun_out/ into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# MVSTER_HEAD=<git sha> in the environment labels the summaries (the box has no .git)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_bench_prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-scene > gpurun_out/r3_bench_line.json 2> gpurun_out/r3_bench_prof.err
echo "stats rc=$?"
timeout -k 10 300 tools/pmc_pass.sh r3_fetch FETCH_SIZE -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_write WRITE_SIZE -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_atomic TCC_EA0_ATOMIC_sum -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
python tools/pmc_summary.py gpurun_out/pmc_r3_fetch gpurun_out/pmc_r3_write gpurun_out/pmc_r3_atomic -o gpurun_out/r3_pmc.json
python tools/pmc_summary.py gpurun_out/pmc_r3_sq -o gpurun_out/r3_pmc_sq.json
python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"
python bench.py --data scene --no-cpu-baseline --no-e2e --no-scene > gpurun_out/r3_bench_final_scene.json 2>/dev/null
