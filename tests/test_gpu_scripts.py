"""-m gpu: the two entry scripts end to end with the reference's command lines (main.py:18-36, evaluate_model.py:17-31) on the
synthetic data set: train one epoch with the fused engine and save (main.py:97-102), resume from that checkpoint (main.py:73-79:
weights AND momentum), train with torch.optim.SGD through autograd from the same checkpoint, evaluate it and write WIDER result
files (evaluate_model.py:60-68, evaluation.py:90-114)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tiny-faces-pytorch_amd")


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


def test_main_and_evaluate_model_scripts(tmp_path):
    main, evalm = os.path.join(PKG, "main.py"), os.path.join(PKG, "evaluate_model.py")
    out = _run([main, "synthetic", "synthetic", "--epochs", "1", "--save-every", "1", "--synthetic-len", "24", "--batch_size", "4", "--lr", "1e-5"], tmp_path)
    assert "Epoch: [0][5/6]" in out and "WARNING: training starts from RANDOM" in out
    ck = tmp_path / "weights" / "checkpoint_1.pth"
    state = torch.load(ck, map_location="cpu")
    assert sorted(state) == ["batch_size", "epoch", "model", "optimizer"] and state["epoch"] == 1 and state["batch_size"] == 4
    assert len(state["model"]) == 571 and len(state["optimizer"]["param_groups"]) == 4
    assert sum("momentum_buffer" in v for v in state["optimizer"]["state"].values()) == 286
    # resume with the fused engine (momentum restored) and with torch.optim.SGD + StepLR (the reference's path)
    out = _run([main, "synthetic", "synthetic", "--epochs", "2", "--save-every", "1", "--synthetic-len", "8", "--batch_size", "4", "--lr", "1e-5",
                "--resume", str(ck)], tmp_path)
    assert "Epoch: [1][1/2]" in out and "Epoch: [0]" not in out
    out = _run([main, "synthetic", "synthetic", "--epochs", "2", "--save-every", "1", "--synthetic-len", "8", "--batch_size", "4", "--lr", "1e-5",
                "--resume", str(ck), "--no-fused", "--dtype", "fp32"], tmp_path)
    assert "Epoch: [1][1/2]" in out
    # evaluation: two synthetic images through the loader contract of evaluate_model.py:60-68, result files of evaluation.py:90-114.
    # (A kaiming-initialised head regresses exp(huge) = inf box sizes, which int() in write_results refuses like the reference's does:
    # evaluate a checkpoint in the same format whose heads are scaled down, bench.py's random-init recipe.)
    sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
    from bench import tame_init_
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_objects=1, num_templates=25)
    tame_init_(m, seed=3)
    with torch.no_grad():                 # untrained scores sit at sigmoid(~0): push the class logits down so that only the tail passes 0.5
        for head in (m.score_res3, m.score_res4):
            head.bias[:25] -= 3.0
    ck = tmp_path / "weights" / "tame.pth"
    torch.save({"epoch": 1, "batch_size": 4, "model": m.state_dict(), "optimizer": {}}, ck)
    out = _run([evalm, "synthetic", "--checkpoint", str(ck), "--num-images", "2", "--prob_thresh", "0.5", "--results_dir", str(tmp_path / "res")], tmp_path)
    files = sorted(os.listdir(tmp_path / "res" / "synthetic"))
    assert files == ["img_0.txt", "img_1.txt"]
    lines = open(tmp_path / "res" / "synthetic" / "img_0.txt").read().split("\n")
    assert lines[0] == "img_0.jpg" and int(lines[1]) == len([l for l in lines[2:] if l.strip()])       # any count, 0 included: the format is the contract
