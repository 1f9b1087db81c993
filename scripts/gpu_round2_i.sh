#!/bin/bash
# round 2, call I: the default bench line (train + forward-pass + single-stream kernels + eval + eval_hard + cpu baseline)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 420 python bench.py > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err
tail -c 4500 gpurun_out/bench_i.json
echo done
