#!/bin/bash
# round 2, call F: kernel-attached fork events (A/B on one box), race screen, timeline
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 600 -k "dual_stream or engine or trainer or grad_ready or slice or wgrad3x3 or train_forward" > gpurun_out/pytest_f.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_f.log
tail -6 gpurun_out/pytest_f.log
for rep in 1 2; do
  for rec in 1 0; do
    if [ $rec = 1 ]; then export TINYFACES_FORK_BY_RECORD=1; else unset TINYFACES_FORK_BY_RECORD; fi
    timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/bench_f_rec${rec}_${rep}.json 2> gpurun_out/bench_f_rec${rec}_${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/bench_f_rec${rec}_${rep}.json"))
print("FORK_BY_RECORD=${rec} rep ${rep}:", d["value"], "img/s", d["ms_per_step"], "ms", [ (k["kernel"], k["ms_per_step"]) for k in d.get("kernels", [])])
PY
  done
done
unset TINYFACES_FORK_BY_RECORD
bash scripts/gpu_trace.sh > gpurun_out/trace_f.log 2>&1
f=$(ls gpurun_out/trace/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f | tee gpurun_out/timeline_f.txt
python scripts/trace_totals.py $f 40 > gpurun_out/trace_totals_f.txt
gzip -c $f > gpurun_out/trace/train_kernel_trace_f.csv.gz
rm -f gpurun_out/trace/*kernel_trace.csv
