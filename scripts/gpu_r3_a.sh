#!/bin/bash
# round 3, call A: contention micro-benchmark + a baseline bench line of the round-2 binary (for A/B on later boxes)
mkdir -p gpurun_out/r3a
timeout 600 python scripts/contention.py > gpurun_out/r3a/contention.txt 2> gpurun_out/r3a/contention.err
echo "contention exit $?"
tail -5 gpurun_out/r3a/contention.err
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-eval-hard > gpurun_out/r3a/bench_base.json 2> gpurun_out/r3a/bench_base.err
echo "bench exit $?"
python -c "
import json; d=json.load(open('gpurun_out/r3a/bench_base.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('eval',{}).get('ms_per_image'))"
