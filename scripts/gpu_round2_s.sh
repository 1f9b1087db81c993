#!/bin/bash
# round 2, call S: full -m gpu suite with conv3x3h + lean wgrad3x3 in the model path; A/B of conv3x3h on layer 2
mkdir -p gpurun_out gpurun_out/s
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 --durations=5 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/s/parity_report_full.txt
for v in 0 256 0 256; do
  export TINYFACES_CONV3H_MINCIN=$v
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/s/bench_$v.json 2> gpurun_out/s/bench_$v.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/s/bench_$v.json').read().strip().splitlines()[-1]); print('mincin $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('forward_pass')['ms'])"
done
