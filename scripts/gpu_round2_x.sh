#!/bin/bash
# round 2, call X: the measurement artefacts of the round with the FINAL binary (conv3x3h, lean wgrad3x3, hardware bf16 conversion), most important first
mkdir -p gpurun_out gpurun_out/final
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
# (1) full GPU suite
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 200 --durations=5 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/final/parity_report_full.txt
# (2) default bench line
timeout 300 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
tail -c 700 gpurun_out/final/bench_n1.json; echo
# (3) rocprofv3 --kernel-trace --stats of the bench command (train)
timeout 150 bash scripts/gpu_prof.sh > gpurun_out/final/prof.log 2>&1; cp gpurun_out/prof_csv/*kernel_stats.csv gpurun_out/final/train_kernel_stats.csv
# (4) HBM traffic: FETCH_SIZE / WRITE_SIZE passes over the bench command
timeout 260 bash scripts/gpu_pmc_bench.sh > gpurun_out/final/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/traffic.json gpurun_out/final/pmc_traffic.json
# (5) per-layer-shape table + timeline of a step
timeout 120 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-eval --layer-table gpurun_out/final/layer_table.json > gpurun_out/final/bench_layer.json 2> gpurun_out/final/bench_layer.err
timeout 150 bash scripts/gpu_trace.sh > gpurun_out/final/trace.log 2>&1
f=$(ls gpurun_out/trace/*kernel_trace.csv | head -1); python scripts/trace_timeline.py $f > gpurun_out/final/timeline.txt; rm -f gpurun_out/trace/*kernel_trace.csv
head -10 gpurun_out/final/timeline.txt
# (6) eval leg kernel stats, smoke
timeout 150 bash scripts/gpu_prof_eval.sh > gpurun_out/final/prof_eval.log 2>&1; cp gpurun_out/prof_eval_csv/*kernel_stats.csv gpurun_out/final/eval_kernel_stats.csv
timeout 120 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
