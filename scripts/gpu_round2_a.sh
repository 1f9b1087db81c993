#!/bin/bash
# round 2, call A: full -m gpu suite (new full-size / hard-setting tests included) + a short bench line
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -c 1500 gpurun_out/bench_a.json
