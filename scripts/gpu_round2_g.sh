#!/bin/bash
# round 2, call G: aux prefetch in the ring-less convs + single fork per layer-3 bottleneck + CU mask knob: parity, then A/B on one box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 600 -k "dual_stream or engine or trainer or grad_ready or slice or train_forward or handover or mask_join or epilogue or real_layer or dgrad" > gpurun_out/pytest_g.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_g.log
tail -6 gpurun_out/pytest_g.log
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/bench_g_$name.json 2> gpurun_out/bench_g_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_g_$name.json"))
print("$name:", d["value"], "img/s", d["ms_per_step"], "ms", [ (k["kernel"], k["ms_per_step"]) for k in d.get("kernels", [])], "fwd", d.get("roofline",{}).get("forward_pass",{}).get("ms"))
PY
}
for rep in 1 2; do
  run base_$rep X=1
  run l3fork_old_$rep TINYFACES_L3_FORK_PER_WGRAD=1
  run cus224_$rep TINYFACES_SIDE_CUS=224
  run cus192_$rep TINYFACES_SIDE_CUS=192
  run cus160_$rep TINYFACES_SIDE_CUS=160
done
