#!/bin/bash
# round 2, call P: per-layer table with conv3x3h active; evaluation legs (configs[1] pyramid, configs[4] 5000-px fp16) with / without it
mkdir -p gpurun_out gpurun_out/p
export PYTHONUNBUFFERED=1
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-eval --layer-table gpurun_out/p/layer_table.json > gpurun_out/p/bench_layer.json 2> gpurun_out/p/bench_layer.err
head -30 gpurun_out/p/layer_table.md
for v in on off; do
  if [ $v = off ]; then export TINYFACES_CONV3H_OFF=1; else unset TINYFACES_CONV3H_OFF; fi
  timeout 400 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/p/bench_eval_$v.json 2> gpurun_out/p/bench_eval_$v.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/p/bench_eval_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], {k: d[k] for k in d if 'eval' in k})"
done
