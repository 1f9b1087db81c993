"""Diagnostic for tests/test_gpu_e2e.py: train the fused engine on the SyntheticFaces images in-process, print the loss trajectory, then look at what the
detector finds on the training images (train-size, one scale) and on the zoomed validation images (reference pyramid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from main import tame_init_
from tinyfaces import ops, transforms, wider_eval
from tinyfaces.datasets.synthetic import SyntheticFaces
from tinyfaces.datasets.templates import load_templates
from tinyfaces.engine import TrainEngine
from tinyfaces.evaluation import get_detections
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel

STEPS = int(os.environ.get("STEPS", "300")); LR = float(os.environ.get("LR", "1e-3")); DT = os.environ.get("DT", "bf16")
templates = load_templates()
BS = int(os.environ.get("BS", "8"))
ds = SyntheticFaces(templates, length=BS * STEPS, seed=0, train=True)
x, cm, rm = [t.cuda() for t in ds.collate([ds[i] for i in range(BS)])]
print("positives per image:", [(cm[i] == 1).sum().item() for i in range(BS)], "faces:", [s[1].shape[0] for s in ds.samples])
torch.manual_seed(0)
m = tame_init_(DetectionModel(num_objects=1, num_templates=25)).set_compute_dtype(DT)
crit = DetectionCriterion(25, seed=0)
crit.ohem_thresh = float(os.environ.get("OHEM", "0.03"))
eng = TrainEngine(m, crit, lr=LR, momentum=0.9, weight_decay=5e-4, device="cuda")
losses = []
DECAY_AT = int(os.environ.get("DECAY_AT", "0"))
for s in range(STEPS):
    if DECAY_AT and s == DECAY_AT:
        eng.set_lr(LR * 0.1)
    x, cm, rm = [t.cuda() for t in ds.collate([ds[BS * s + i] for i in range(BS)])]
    losses.append(eng.step(x, cm, rm))
torch.cuda.synchronize()
L = np.array([[float(v) for v in l] for l in losses]) / BS
for w in range(0, STEPS, max(1, STEPS // 15)):
    print(f"steps {w:4d}..: cls {L[w:w + STEPS // 15, 0].mean():9.3f} reg {L[w:w + STEPS // 15, 1].mean():9.3f}")
print("last 50 steps: cls %.3f reg %.4f" % (L[-50:, 0].mean(), L[-50:, 1].mean()))
x, cm, rm = [t.cuda() for t in ds.collate([ds[i] for i in range(BS)])]       # fresh (unmined) targets of the first batch
print("target values:", {float(v): int((cm == v).sum()) for v in cm.unique()})
with torch.no_grad():
    y_tr = m._run_forward(x, training=True).float()
    m.eval()
    y_ev = m(x).float()
for nm, y in (("train-mode", y_tr), ("eval-mode", y_ev)):
    c = y[:, :25]
    pos, neg = cm == 1, cm == -1
    print(f"{nm}: positives with logit > 0: {float((c[pos] > 0).float().mean()):.3f} (mean logit {float(c[pos].mean()):.2f}); negatives with logit < 0: {float((c[neg] < 0).float().mean()):.5f} "
          f"(max {float(c[neg].max()) if neg.any() else 0:.2f}); ignored zone: {int((cm == 0).sum())} entries, fraction > 0: {float((c[cm == 0] > 0).float().mean()) if (cm == 0).any() else 0:.4f}")
print("train vs eval logits: max abs diff", float((y_tr - y_ev).abs().max()), "mean", float((y_tr - y_ev).abs().mean()))
eng.close()
m.eval()
tfm = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
dv = SyntheticFaces(templates, length=8, seed=0, train=False)
gt = dv.ground_truth()
def iou(a, b):
    x1, y1 = max(a[0], b[0]), max(a[1], b[1]); x2, y2 = min(a[2], b[2]), min(a[3], b[3])
    i = max(0, x2 - x1) * max(0, y2 - y1)
    return i / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i)
for scales in ((0,), (-2, -1, 0, 1)):
    preds = {"faces": {}}
    for i in range(8):
        img, name = dv[i]
        d = get_detections(m, img, templates, ops.RF, tfm, prob_thresh=float(os.environ.get("THR", "0.1")), nms_thresh=0.3, scales=scales, device="cuda", pyramid_on_gpu=True,
                           mask_axis=os.environ.get("MASK_AXIS", "template"))
        g = gt[f"img_{i}"]; gb = np.column_stack([g[:, 0], g[:, 1], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]])
        order = np.argsort(-d[:, 4]) if d.shape[0] else []
        d = d[order] if d.shape[0] else d
        best = [max((iou(b, q) for q in d[:, :4]), default=0) for b in gb]
        top = [(round(float(r[4]), 2), round(max(iou(r[:4], b) for b in gb), 2)) for r in d[:6]]
        print(f"scales {scales} img {i}: {d.shape[0]:4d} detections; best IoU per face {np.round(best, 2)}; top (score, IoU) {top}")
        rows = np.column_stack([d[:, 0], d[:, 1], d[:, 2] - d[:, 0] + 1, d[:, 3] - d[:, 1] + 1, d[:, 4]]) if d.shape[0] else np.zeros((0, 5))
        preds["faces"][f"img_{i}"] = rows
    keep = {"faces": {k: np.arange(v.shape[0]) for k, v in gt.items()}}
    ap, _ = wider_eval.evaluate_setting(wider_eval.norm_scores(preds), {"faces": gt}, keep)
    print(f"scales {scales}: AP = {ap:.4f}")
# what exactly comes out for image 0 at the training scale
img, _ = dv[0]
d, cand, keep = get_detections(m, img, templates, ops.RF, tfm, prob_thresh=0.5, nms_thresh=0.3, scales=(0,), device="cuda", pyramid_on_gpu=True, mask_axis="template", return_candidates=True)
g = gt["img_0"]; print("GT (x1 y1 x2 y2):", np.round(np.column_stack([g[:, 0], g[:, 1], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]]), 0).tolist())
print("candidates:", cand.shape[0], "kept:", d.shape[0])
for r in d[np.argsort(-d[:, 4])][:12]:
    print("   det", np.round(r[:4], 0).tolist(), "w x h", round(r[2] - r[0]), round(r[3] - r[1]), "score", round(float(r[4]), 2))
# the training targets of the same image: positives per template
xb, cmb, rmb = [t.cuda() for t in ds.collate([ds[0]])]
pos = torch.nonzero(cmb[0] == 1)
print("training positives of view 0 (template, fy, fx):", pos[:20].tolist(), "of", pos.shape[0])
with torch.no_grad():
    y1 = m(xb).float()[0]
print("eval logits at those positives:", [round(float(y1[t, fy, fx]), 1) for t, fy, fx in pos[:20].tolist()])
top = torch.topk(y1[:25].flatten(), 10)
print("top-10 logits anywhere (template, fy, fx, logit, target):", [(int(i // (63 * 63)), int(i % (63 * 63) // 63), int(i % 63), round(float(v), 1), float(cmb[0].flatten()[i])) for v, i in zip(top.values, top.indices)])
