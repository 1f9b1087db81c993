"""The drop-in boundary, exercised the way the reference's own callers use it: the STATEMENTS of
`/root/reference/evaluate_model.py:48-68` (`run`) and `/root/reference/main.py:67-102` (optimizer / resume / checkpoint) are
restated here against the product's loader, model and engine.  Host only: nothing below touches a GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch


def _wider_tree(root, split, names, size=(40, 50)):
    from PIL import Image
    rng = np.random.RandomState(3)
    for n in names:
        d = root / f"WIDER_{split}" / "images" / os.path.dirname(n)
        d.mkdir(parents=True, exist_ok=True)
        Image.fromarray(rng.randint(0, 256, size + (3,), dtype=np.uint8), "RGB").save(root / f"WIDER_{split}" / "images" / n)


def test_reference_eval_loop_statements_hold_on_the_product_loader(tmp_path):
    """evaluate_model.py:60-68: `for _, (img, filename) in enumerate(val_loader)`, `img[0]`, `filename[0]`, `len(val_loader)`,
    `val_loader.dataset.rf`, `val_loader.dataset.transforms`."""
    from tinyfaces import transforms
    from tinyfaces.datasets import get_dataloader
    names = ["3--Riot/a.png", "3--Riot/b.png", "5--Car/c.png"]
    _wider_tree(tmp_path, "val", names)
    (tmp_path / "val.txt").write_text("".join(f"{n}\n1\n4 5 10 12 0 0 0 0 0 0\n" for n in names))
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    args = SimpleNamespace(batch_size=1, workers=0, dataset_root=str(tmp_path), debug=False)
    val_loader, templates = get_dataloader(tmp_path / "val.txt", args, train=False, split="val", img_transforms=tf)
    assert templates.shape == (25, 5) and len(val_loader) == 3 and val_loader.batch_size == 1
    seen = []
    for _, (img, filename) in enumerate(val_loader):
        one = img[0]                                           # what get_detections receives (evaluate_model.py:61)
        assert one.dim() == 3 and one.shape[0] == 3 and one.dtype == torch.float32 and 0.0 <= float(one.min()) and float(one.max()) <= 1.0
        assert isinstance(filename[0], str) and filename[0].endswith(".png") and "/" in filename[0]     # write_results' img_path
        seen.append(filename[0])
    assert seen == names                                       # val is not shuffled (datasets/__init__.py:48: shuffle=train)
    assert val_loader.dataset.rf == {"size": [859, 859], "stride": [8, 8], "offset": [-1, -1]}
    assert val_loader.dataset.transforms is tf


def test_train_loader_answers_what_main_py_asks(tmp_path):
    """main.py:99 `train_loader.batch_size`; trainer.py:89 `len(dataloader)`; the synthetic loader too."""
    from tinyfaces.datasets import get_dataloader
    names = [f"0--P/{i}.png" for i in range(5)]
    _wider_tree(tmp_path, "train", names)
    (tmp_path / "train.txt").write_text("".join(f"{n}\n1\n4 5 10 12 0 0 0 0 0 0\n" for n in names))
    args = SimpleNamespace(batch_size=2, workers=0, dataset_root=str(tmp_path), debug=False)
    loader, _ = get_dataloader(tmp_path / "train.txt", args)
    assert loader.batch_size == 2 and len(loader) == 3 and len(loader.dataset) == 5
    loader.set_epoch(1)                                        # single process: a no-op that must exist
    syn, _ = get_dataloader("synthetic", SimpleNamespace(batch_size=12, synthetic_len=48))
    assert syn.batch_size == 12 and len(syn) == 4
    val, _ = get_dataloader("synthetic", SimpleNamespace(batch_size=1, synthetic_len=2), train=False, split="val")
    (img, filename), _ = list(val)
    assert img.shape == (1, 3, 960, 1280) and filename == ["synthetic/img_0.jpg"]


def _engine_on_cpu():
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel
    torch.manual_seed(0)
    m = DetectionModel(num_objects=1, num_templates=25)
    return TrainEngine(m, DetectionCriterion(25), lr=1e-4, momentum=0.9, weight_decay=5e-4, device="cpu")


def test_fused_checkpoint_is_a_torch_sgd_state_dict_and_round_trips(tmp_path):
    """The reference's resume path (main.py:67-83): `optimizer = optim.SGD(model.learnable_parameters(lr), ...)`,
    `optimizer.load_state_dict(checkpoint['optimizer'])`, `StepLR(optimizer, 20, last_epoch=start_epoch - 1)` must accept a
    checkpoint written by the fused engine, momentum included; and the engine must resume from it (and from a checkpoint
    whose optimizer entry is a real torch.optim.SGD state_dict) with the same momentum buffers."""
    from torch import optim
    from tinyfaces import trainer
    from tinyfaces.evaluation import get_model
    from tinyfaces.models.model import DetectionModel
    eng = _engine_on_cpu()
    g = torch.Generator().manual_seed(1)
    eng.flat_m.copy_(torch.randn(eng.flat_m.shape, generator=g))
    eng.steps = 3
    snapshot = {"epoch": 10, "batch_size": 12, "model": eng.model.state_dict(), "optimizer": eng.optimizer_state_dict(base_lr=1e-4)}
    trainer.save_checkpoint(snapshot, filename="checkpoint_10.pth", save_path=tmp_path / "weights")
    ckpt = torch.load(tmp_path / "weights" / "checkpoint_10.pth", map_location="cpu")
    assert sorted(ckpt) == ["batch_size", "epoch", "model", "optimizer"]                     # main.py:97-102

    # ---- the reference's statements
    model = DetectionModel(num_objects=1, num_templates=25)
    optimizer = optim.SGD(model.learnable_parameters(1e-4), lr=1e-4, momentum=0.9, weight_decay=0.0005)
    model.load_state_dict(ckpt["model"])
    optimizer.load_state_dict(ckpt["optimizer"])
    scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=20, last_epoch=ckpt["epoch"] - 1)
    assert [pg["lr"] for pg in optimizer.param_groups] == pytest.approx([1e-4, 1e-5, 1e-4, 0.0])
    assert [len(pg["params"]) for pg in optimizer.param_groups] == [284, 2, 2, 1]
    name_of = {id(p): k for k, p in model.named_parameters()}
    n_buf = 0
    for pg in optimizer.param_groups:
        for p in pg["params"]:
            st = optimizer.state.get(p, {})
            if "momentum_buffer" in st:
                o, n = eng.model._segments[name_of[id(p)]]
                assert torch.equal(st["momentum_buffer"].reshape(-1), eng.flat_m[o:o + n])
                n_buf += 1
    assert n_buf == 284 - 2 + 2 + 2                          # every trained tensor; not model.fc.* (no gradient), not the lr-0 upsample
    # one torch step from the loaded state works (fc / upsample have no grad)
    for p in model.parameters():
        p.grad = None
    model.score_res3.bias.grad = torch.ones_like(model.score_res3.bias)
    optimizer.step(); scheduler.step()

    # ---- and back: the engine resumes from a torch.optim.SGD state_dict (the reference's own checkpoints)
    ckpt = torch.load(tmp_path / "weights" / "checkpoint_10.pth", map_location="cpu")    # (load_state_dict shares tensors: the step above mutated them)
    eng2 = _engine_on_cpu()
    assert float(eng2.flat_m.abs().sum()) == 0.0
    opt_ref = optim.SGD(eng.model.learnable_parameters(1e-4), lr=1e-4, momentum=0.9, weight_decay=0.0005)
    opt_ref.load_state_dict(ckpt["optimizer"])
    assert eng2.load_optimizer_state_dict(opt_ref.state_dict())
    for k, (o, n) in eng.model._segments.items():             # the lr-0 upsample weight carries no state (its slot stays zero)
        want = torch.zeros(n) if k == "score4_upsample.weight" else eng.flat_m[o:o + n]
        assert torch.equal(eng2.flat_m[o:o + n], want), k
    assert eng2.load_optimizer_state_dict({}) is False        # a round-1 checkpoint ("optimizer": {}) resumes without momentum

    # ---- save -> get_model round trip (evaluation.py:12-17)
    loaded = get_model(str(tmp_path / "weights" / "checkpoint_10.pth"), num_templates=25)
    a, b = eng.model.state_dict(), loaded.state_dict()
    assert list(a) == list(b) and len(a) == 571
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_pretrained_trunk_loads_a_torchvision_style_state_dict(tmp_path):
    """main.py --pretrained / DetectionModel(pretrained_weights=<path>): a bare resnet101 state_dict (keys without the `model.`
    prefix, `layer4.*` present) lands in the trunk; a non-path value warns loudly instead of being ignored silently."""
    from tinyfaces.models.model import DetectionModel, resnet101
    torch.manual_seed(5)
    trunk = resnet101()
    sd = {k: v.clone() for k, v in trunk.state_dict().items()}
    sd["layer4.0.conv1.weight"] = torch.zeros(512, 1024, 1, 1)          # torchvision files carry layer4: dropped like model.py:23
    torch.save(sd, tmp_path / "resnet101.pth")
    m = DetectionModel(pretrained_weights=str(tmp_path / "resnet101.pth"), num_templates=25)
    assert torch.equal(m.model.layer3[22].conv3.weight, trunk.layer3[22].conv3.weight)
    assert torch.equal(m.model.fc.weight, trunk.fc.weight)
    with pytest.warns(RuntimeWarning, match="RANDOM"):
        DetectionModel(pretrained_weights="ResNet101_Weights.IMAGENET1K_V1" if False else object(), num_templates=1)
