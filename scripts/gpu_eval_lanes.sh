#!/bin/bash
# A/B of the evaluation pyramid over the lane count / enqueue order of DetectionModel.forward_levels (bench.py --eval-only, configs[1])
R=${GRAFT_REPO_ROOT:-$(pwd)}
out="$R/gpurun_out/eval_lanes.txt"; : > "$out"
for rep in 1 2 3; do
  for cfg in "1 big" "2 big" "3 big" "2 small" "3 small"; do
    set -- $cfg
    ms=$(TINYFACES_EVAL_LANES=$1 TINYFACES_EVAL_LANES_ORDER=$2 python bench.py --eval-only 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['eval']['ms_per_image'])")
    echo "[$rep] lanes=$1 order=$2 -> $ms ms/image" | tee -a "$out"
  done
done
