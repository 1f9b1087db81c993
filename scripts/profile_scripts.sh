#!/bin/bash
# cProfile of the two bundled entry scripts on the GPU box (host-side overheads of the product loops):  bash scripts/profile_scripts.sh [train|eval]
R=${GRAFT_REPO_ROOT:-$(pwd)}; export PYTHONPATH=$R/tiny-faces-pytorch_amd:$R
cd /tmp
if [ "${1:-eval}" = train ]; then
  python -m cProfile -s cumtime $R/tiny-faces-pytorch_amd/main.py synthetic-faces synthetic-faces --epochs 1 --synthetic-len 3200 --batch_size 32 --lr 2e-4 --save-every 9 --save-path /tmp/w --dtype bf16 --ohem-thresh 0 2>&1 | grep -v "^Epoch\|it/s" | sed -n 3,30p | cut -c1-160
else
  python - <<PY
import sys, torch
sys.path[:0] = ["$R", "$R/tiny-faces-pytorch_amd"]
from bench import tame_init_
from tinyfaces.models.model import DetectionModel
m = DetectionModel(num_objects=1, num_templates=25); tame_init_(m, seed=3)
with torch.no_grad():
    for h in (m.score_res3, m.score_res4): h.bias[:25] -= 3.0
torch.save({"epoch": 1, "batch_size": 4, "model": m.state_dict(), "optimizer": {}}, "/tmp/tame.pth")
PY
  python -m cProfile -s cumtime $R/tiny-faces-pytorch_amd/evaluate_model.py synthetic --checkpoint /tmp/tame.pth --num-images ${N:-64} --prob_thresh 0.5 --results_dir /tmp/res 2>&1 | grep -v "it/s\|detections" | sed -n 1,40p | cut -c1-160
fi
