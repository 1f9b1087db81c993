#!/bin/bash
# round 2, call K: full -m gpu suite with the final binary (parity report), smoke, per-layer-shape table of the timed region
mkdir -p gpurun_out gpurun_out/final
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 --durations=5 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/final/parity_report_full.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-eval --layer-table gpurun_out/final/layer_table.json > gpurun_out/final/bench_layer.json 2> gpurun_out/final/bench_layer.err
head -8 gpurun_out/final/layer_table.md
