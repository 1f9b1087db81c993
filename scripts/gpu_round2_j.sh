#!/bin/bash
# round 2, call J: the measurement artefacts of the round with the final binary
mkdir -p gpurun_out gpurun_out/final
export PYTHONUNBUFFERED=1
# (1) default bench line
timeout 420 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
tail -c 1200 gpurun_out/final/bench_n1.json
# (2) rocprofv3 --kernel-trace --stats of the bench command (train) and of the eval leg
bash scripts/gpu_prof.sh > gpurun_out/final/prof.log 2>&1; cp gpurun_out/prof_csv/*kernel_stats.csv gpurun_out/final/train_kernel_stats.csv
bash scripts/gpu_prof_eval.sh > gpurun_out/final/prof_eval.log 2>&1; cp gpurun_out/prof_eval_csv/*kernel_stats.csv gpurun_out/final/eval_kernel_stats.csv
# (3) HBM traffic: FETCH_SIZE / WRITE_SIZE passes over the bench command
bash scripts/gpu_pmc_bench.sh > gpurun_out/final/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/traffic.json gpurun_out/final/pmc_traffic.json
# (4) SQ counters of the conv tile candidates and the 3x3 weight-gradient kernels
TAG=r2 SCRIPT=scripts/pmc_kernels_r2.py bash scripts/gpu_pmc.sh > gpurun_out/final/pmc_sq.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc/r2_sq.csv "conv_dma|wgrad" > gpurun_out/final/pmc_sq_summary.txt; cat gpurun_out/final/pmc_sq_summary.txt
# (5) per-layer-shape table + timeline of a step
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-eval --layer-table gpurun_out/final/layer_table.json > gpurun_out/final/bench_layer.json 2> gpurun_out/final/bench_layer.err
bash scripts/gpu_trace.sh > gpurun_out/final/trace.log 2>&1
f=$(ls gpurun_out/trace/*kernel_trace.csv | head -1); python scripts/trace_timeline.py $f > gpurun_out/final/timeline.txt; rm -f gpurun_out/trace/*kernel_trace.csv
head -12 gpurun_out/final/timeline.txt
# (6) the N > 1 code path on this 1-GPU box (RCCL with a 1-rank group; 2 gloo ranks sharing the GPU)
bash scripts/gpu_dist_smoke.sh > gpurun_out/final/dist_smoke.log 2>&1; tail -6 gpurun_out/final/dist_smoke.log
# (7) smoke
timeout 300 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
