#!/bin/bash
# Round-3 profile collection on the GPU box (gpurun): kernel stats, FETCH / WRITE / ATOMIC / SQ counter passes (separate runs,
# kernel-trace only), the un-profiled bench lines, the end-to-end step windows, the convolution benchmark and its matrix-core
# occupancy.  Copy the summaries you want judged from gpurun_out/ into profiles/ (tools/collect_profiles.sh part: hot | e2e).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
part=${1:-hot}
# MVSTER_HEAD=<git sha> in the environment labels the summaries (the box has no .git)
if [ $part = hot ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_bench_prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-scene > gpurun_out/r3_bench_line.json 2> gpurun_out/r3_bench_prof.err
echo "stats rc=$?"
cp gpurun_out/r3_bench_prof/*/*kernel_stats.csv gpurun_out/r3_bench_kernel_stats.csv
timeout -k 10 300 tools/pmc_pass.sh r3_fetch FETCH_SIZE -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_write WRITE_SIZE -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_atomic TCC_EA0_ATOMIC_sum -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
timeout -k 10 300 tools/pmc_pass.sh r3_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-scene
python tools/pmc_summary.py gpurun_out/pmc_r3_fetch gpurun_out/pmc_r3_write gpurun_out/pmc_r3_atomic -o gpurun_out/r3_pmc.json
python tools/pmc_summary.py gpurun_out/pmc_r3_sq -o gpurun_out/r3_pmc_sq.json
cp gpurun_out/r3_pmc.json profiles/r3_pmc.json            # (the box's scratch copy: bench.py then reports roofline.traffic)
python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$?"
python bench.py --data scene --no-cpu-baseline --no-e2e --no-scene > gpurun_out/r3_bench_final_scene.json 2>/dev/null
fi
if [ $part = e2e ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_e2e_prof -- python3 tools/e2e_bench.py --steps 3 > gpurun_out/r3_e2e_prof.log 2>&1
cp gpurun_out/r3_e2e_prof/*/*kernel_stats.csv gpurun_out/r3_e2e_kernel_stats.csv
python tools/step_window.py gpurun_out/r3_e2e_prof/*/*kernel_trace.csv --ms 41 --top 60 > gpurun_out/r3_e2e_step_window.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_e2e_eval_prof -- python3 tools/e2e_bench.py --steps 3 --eval > gpurun_out/r3_e2e_eval_prof.log 2>&1
python tools/step_window.py gpurun_out/r3_e2e_eval_prof/*/*kernel_trace.csv --ms 9.6 --top 40 > gpurun_out/r3_e2e_eval_step_window.txt
for a in "--graph" "" "--graph --eval" "--eval" "--graph --batch 4"; do python tools/e2e_bench.py --steps 8 $a 2>&1 | grep "^e2e"; done > gpurun_out/r3_e2e_bench.txt
python tools/conv_bench.py --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_conv_bench.txt
timeout -k 10 300 tools/pmc_pass.sh r3_conv_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU -- tools/conv_bench.py --reps 2
python tools/mfma_occupancy.py gpurun_out/pmc_r3_conv_mfma > gpurun_out/r3_conv_mfma_occupancy.txt
fi
