#!/bin/bash
# round 2, call Q: lean wgrad3x3 (precomputed ring offsets, branch-free DMA walk, staged 16-byte partial stores): tests, traced microbench, ablations, step
mkdir -p gpurun_out gpurun_out/q
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 300 -k "wgrad" > gpurun_out/q/pytest.log 2>&1
echo "pytest exit $?: $(tail -1 gpurun_out/q/pytest.log)"
TAG=r2q PAT="wgrad" bash scripts/gpu_ubench_trace.sh scripts/microbench_wgrad3.py
tail -4 gpurun_out/ub/r2q.log
cd $GRAFT_REPO_ROOT
for d in 1 2 3; do
  TINYFACES_WGRAD3_DBG=$d TAG=r2q_dbg$d PAT="wgrad3x3_kernel" bash scripts/gpu_ubench_trace.sh scripts/microbench_wgrad3.py > /dev/null
  echo "== dbg $d"; grep "wgrad3x3_kernel" gpurun_out/ub/r2q_dbg$d.txt
  cd $GRAFT_REPO_ROOT
done
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err
python -c "import json,sys; d=json.loads(open('gpurun_out/q/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernels'])"
