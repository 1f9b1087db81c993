#!/bin/bash
# round 2, call Y: DPP / permlane-swap reductions in the conv epilogues and wave_sum: the tests that exercise them, A/B on the step; the fixed script test
mkdir -p gpurun_out gpurun_out/y
export PYTHONUNBUFFERED=1
timeout 280 python -m pytest tests/test_gpu_conv.py tests/test_gpu_small_ops.py tests/test_gpu_model.py tests/test_gpu_scripts.py -q -m gpu -p no:cacheprovider --timeout 200 -x > gpurun_out/y/pytest.log 2>&1
echo "pytest exit $?: $(tail -1 gpurun_out/y/pytest.log)"
for v in new old new old; do
  if [ $v = old ]; then export TF_CONV_DBG=16; else unset TF_CONV_DBG; fi
  timeout 120 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/y/bench_$v.json 2> gpurun_out/y/bench_$v.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/y/bench_$v.json').read().strip().splitlines()[-1]); print('shfl $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('forward_pass')['ms'])"
done
