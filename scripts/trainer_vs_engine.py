"""img/s of the reference-compatible step (trainer.train's body: autograd + torch.optim.SGD, tinyfaces/trainer.py) against the fused TrainEngine on the
bench batch (bs = 12, 500 x 500, bf16)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from bench import tame_init_, synthetic_batch
from tinyfaces import ops
from tinyfaces.datasets.templates import load_templates
from tinyfaces.engine import TrainEngine
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel
dev = torch.device("cuda")
templates = load_templates()
t_d = torch.as_tensor(templates, dtype=torch.float64, device=dev)
b = synthetic_batch(0, 12, dev, t_d)
cm, rm = ops.dense_overlap_targets_device(b["boxes"], b["offs"], b["total"], t_d, paste_d=b["paste"], seed=0)
x = b["x"]
def run(kind, steps=15, warm=4):
    torch.manual_seed(0)
    m = tame_init_(DetectionModel(num_objects=1, num_templates=25)).set_compute_dtype("bf16").to(dev).train()
    crit = DetectionCriterion(25, seed=0, lazy_meters=True)
    if kind == "engine":
        eng = TrainEngine(m, crit, lr=1e-4, momentum=0.9, weight_decay=5e-4, device=dev)
        step = lambda: eng.step(x, cm.clone(), rm)
    else:
        opt = torch.optim.SGD(m.learnable_parameters(1e-4), lr=1e-4, momentum=0.9, weight_decay=5e-4)
        def step():
            loss = crit(m(x), cm.clone(), rm)
            opt.zero_grad()
            loss.backward()
            opt.step()
    for _ in range(warm): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    crit._pending.clear()
    print(f"{kind:8s} {dt * 1e3:7.2f} ms/step  {12 / dt:7.1f} img/s", flush=True)
run("engine"); run("trainer")
