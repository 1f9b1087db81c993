#!/bin/bash
# A/B of bench.py variants on ONE box (boxes differ by a few %): each argument is "ENV=.. ENV=.." applied to one run; 2 rounds
for round in 1 2; do
  for v in "$@"; do
    r=$(env $v python bench.py --steps ${STEPS:-15} --warmup 3 --no-cpu-baseline --no-eval --no-profile --no-fp32-path 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'loss', d['loss']['cls'], d['loss']['reg'])")
    echo "[$round] $v -> $r"
  done
done
