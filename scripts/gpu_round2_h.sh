#!/bin/bash
# round 2, call H: full -m gpu suite with every change of the round + bench (train, eval, eval_hard, cpu baseline)
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_h.json 2> gpurun_out/bench_h.err
tail -c 3500 gpurun_out/bench_h.json
tail -5 gpurun_out/bench_h.err
