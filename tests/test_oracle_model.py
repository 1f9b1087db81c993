"""CPU: oracle network / pyramid / trainer restatements vs goldens produced by running the
reference's own model.py / evaluation.py / trainer.py (over the stubbed torchvision trunk)."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from oracle import refstub
from oracle.criterion import criterion as oracle_criterion
from oracle.model import OracleDetectionModel, bilinear_kernel, tame_init_
from oracle.pyramid import get_detections
from oracle.targets import RF


@pytest.fixture(scope="module")
def omodel():
    return tame_init_(OracleDetectionModel(num_templates=25), 0)


def test_state_dict_contract(golden, omodel):
    g = golden("model")
    sd = omodel.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"].tolist()
    assert [v.numel() for v in sd.values()] == g["state_dict_numel"].tolist()
    assert len(sd) == 571
    assert sum(p.numel() for p in omodel.parameters()) == 30026674
    assert np.array_equal(bilinear_kernel(4).astype(np.float32), g["bilinear_diag"])
    assert float(g["bilinear_offdiag_abs_sum"]) == 0.0
    lp = omodel.learnable_parameters(0.5)
    assert [len(list(x["params"])) for x in lp] == g["group_sizes"].tolist() == [284, 2, 2, 1]
    assert [x["lr"] for x in lp] == g["group_lrs"].tolist()


@pytest.mark.parametrize("ci", range(4))
def test_forward_backward_vs_reference(golden, ci):
    g = golden("model")
    tag = f"m{ci}"
    m = tame_init_(OracleDetectionModel(num_templates=25), 0)
    m.train(bool(g[f"{tag}_train"]))
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    y = m(x)
    assert y.shape == g[f"{tag}_y"].shape
    assert np.allclose(y.detach().numpy(), g[f"{tag}_y"], rtol=1e-5, atol=1e-6)
    y.backward(torch.from_numpy(g[f"{tag}_gy"]))
    assert np.allclose(x.grad.numpy(), g[f"{tag}_gx"], rtol=1e-4, atol=1e-6)
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(f"{tag}_grad::"):
            ref = g[k]
            assert np.allclose(params[k.split("::")[1]].grad.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max()), k
        if k.startswith(f"{tag}_rm::") or k.startswith(f"{tag}_rv::"):
            assert np.allclose(m.state_dict()[k.split("::")[1]].numpy(), g[k], rtol=1e-5, atol=1e-6), k


def test_pyramid_detections_vs_reference(golden, omodel):
    g = golden("detections")
    templates = golden("targets")["templates"]
    tf = refstub.Compose([refstub.ToTensor(), refstub.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    dets = get_detections(omodel, torch.from_numpy(g["img"]), templates, RF, tf, prob_thresh=float(g["thr"]),
                          nms_thresh=0.3, scales=tuple(g["scales"].tolist()))
    ref = g["dets_ref_K4"]
    assert dets.shape == (ref.shape[0], 5) and ref.shape[0] > 100
    assert np.allclose(dets[:, :4], ref, rtol=1e-6, atol=1e-6)     # same rows, same order (D2: + score column)


def test_trainer_two_steps_vs_reference(golden):
    """Restated trainer.train loop (trainer.py:68-90) on the oracle model."""
    g = golden("trainer")
    m = tame_init_(OracleDetectionModel(num_templates=25), 0)
    m.train()
    opt = torch.optim.SGD(m.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    np.random.seed(3)
    n_avg, cls_avg, reg_avg = 0, 0.0, 0.0
    lines = []
    for i in range(2):
        x = torch.from_numpy(g[f"b{i}_img"])
        cm = torch.from_numpy(g[f"b{i}_cm"].astype(np.float32))
        rm = torch.from_numpy(g[f"b{i}_rm"]).float()
        out = m(x)
        r = oracle_criterion(out, cm, rm)
        opt.zero_grad()
        out.backward(r["grad"])
        opt.step()
        cls_avg = (n_avg * cls_avg + r["cls"]) / (n_avg + 2)
        reg_avg = (n_avg * reg_avg + r["reg"]) / (n_avg + 2)
        n_avg += 2
        lines.append("Epoch: [0][{}/2]\t\tloss_cls: {:.6f}\tloss_reg: {:.6f}".format(i, cls_avg, reg_avg))
    ref_lines = str(g["log"]).strip().split("\n")
    for a, b in zip(lines, ref_lines):
        va = [float(t.split(": ")[1]) for t in a.split("\t") if "loss" in t]
        vb = [float(t.split(": ")[1]) for t in b.split("\t") if "loss" in t]
        assert np.allclose(va, vb, rtol=1e-4), (a, b)
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("after::"):
            assert np.allclose(sd[k.split("::")[1]].numpy(), g[k], rtol=1e-4, atol=1e-6), k
    chk = float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(chk - float(g["param_checksum"])) < 1e-2 * max(1.0, abs(chk)) * 1e-3
