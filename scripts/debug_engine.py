import os, sys, io
import numpy as np, torch
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd"), os.path.join(ROOT, "tests")]
from oracle.model import OracleDetectionModel, tame_init_
from tinyfaces import trainer
from tinyfaces.engine import TrainEngine
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel
g = np.load(os.path.join(ROOT, "tests/golden/trainer.npz"))
batches = [(torch.from_numpy(g[f"b{i}_img"]), torch.from_numpy(g[f"b{i}_cm"].astype(np.float32)), torch.from_numpy(g[f"b{i}_rm"]).float()) for i in range(2)]
E = 25 * 16 * 16
keep = torch.ones(2, E, dtype=torch.uint8); keep[:, 128:] = 0
om = tame_init_(OracleDetectionModel(num_templates=25), 0)
def fresh():
    m = DetectionModel(num_templates=25); m.load_state_dict(om.state_dict()); m.set_compute_dtype(torch.float32)
    c = DetectionCriterion(25); c.inject_sampling(keep, keep); return m, c
def diff(a, b, tag):
    rows = []
    for k in a:
        if a[k].is_floating_point():
            rows.append((float((a[k].float().cpu() - b[k].float().cpu()).abs().max() / (b[k].float().abs().max().cpu() + 1e-30)), k))
    rows.sort(reverse=True)
    print(tag, rows[:4])
for nsteps in (1, 2):
    m1, c1 = fresh()
    opt = torch.optim.SGD(m1.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    with redirect_stdout(io.StringIO()):
        trainer.train(m1, c1, opt, batches[:nsteps], 0, torch.device("cuda"))
    m2, c2 = fresh()
    eng = TrainEngine(m2, c2, lr=1e-3, momentum=0.9, weight_decay=5e-4, device="cuda")
    for img, cm, rm in batches[:nsteps]:
        eng.step(img.cuda(), cm.cuda(), rm.cuda())
    diff(m2.state_dict(), m1.state_dict(), f"engine vs trainer after {nsteps} step(s):")
    # trainer vs trainer (run-to-run)
    m3, c3 = fresh()
    opt3 = torch.optim.SGD(m3.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    with redirect_stdout(io.StringIO()):
        trainer.train(m3, c3, opt3, batches[:nsteps], 0, torch.device("cuda"))
    diff(m3.state_dict(), m1.state_dict(), f"trainer vs trainer after {nsteps} step(s):")
# gradient itself, engine path vs autograd path, step 1
m1, c1 = fresh(); m1 = m1.cuda().train()
img, cm, rm = batches[0]
out = m1(img.cuda()); loss = c1(out, cm.cuda(), rm.cuda()); loss.backward()
ga = {k: p.grad.clone() for k, p in m1.named_parameters() if p.grad is not None}
m2, c2 = fresh(); eng = TrainEngine(m2, c2, lr=0.0, momentum=0.0, weight_decay=0.0, device="cuda")
eng.step(img.cuda(), cm.cuda(), rm.cuda())
gf = m2._grad_flat_persistent
rows = []
for k, (o, n) in m2._segments.items():
    a = gf[o:o + n].view_as(ga[k])
    rows.append((float((a - ga[k]).abs().max() / (ga[k].abs().max() + 1e-30)), k))
rows.sort(reverse=True); print("grad engine vs autograd:", rows[:4])
