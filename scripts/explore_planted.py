"""Exploration for the bf16 / fp16 detection-parity fixture: a planted image (grey background + a few high-contrast squares) through
the random-init detector.  Prints, for the fp32 GPU path, the sorted top logits, the widest gaps, and the bf16 / fp16 error at the candidates."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from bench import tame_init_
from tinyfaces.models.model import DetectionModel
from tinyfaces.datasets.templates import load_templates

def planted(seed=0, H=960, W=1280, k=8, size=48):
    rs = np.random.RandomState(seed)
    img = np.full((H, W, 3), 0.45, np.float32)
    cells = [(r, c) for r in range(1, 5) for c in range(1, 7)]
    rs.shuffle(cells)
    for i, (r, c) in enumerate(cells[:k]):
        y, x = r * 192 - 24, c * 192 - 64
        s = size + 16 * (i % 3)
        patch = rs.rand(s // 8, s // 8, 3).astype(np.float32)
        img[y:y + s, x:x + s] = np.kron(patch, np.ones((8, 8, 1), np.float32))
    return torch.from_numpy(img).permute(2, 0, 1).contiguous()

templates = load_templates()
torch.manual_seed(0)
base = tame_init_(DetectionModel(num_objects=1, num_templates=25))
sd = base.state_dict()
img = planted()
mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
outs = {}
for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    m = DetectionModel(num_objects=1, num_templates=25); m.load_state_dict(sd); m = m.cuda().set_compute_dtype(dt).eval()
    lv = []
    with torch.no_grad():
        for s in (0.5, 1, 2):
            x = torch.nn.functional.interpolate(img.unsqueeze(0), scale_factor=s, mode="bilinear", align_corners=False) if s != 1 else img.unsqueeze(0)
            x = ((x - mean) / std).cuda()
            lv.append(m(x)[0, :25].float().cpu().numpy())
    outs[name] = lv
for li, s in enumerate((0.5, 1, 2)):
    a = outs["fp32"][li]
    flat = np.sort(a.ravel())[::-1]
    print(f"level x{s}: map {a.shape} logits min {a.min():.4f} median {np.median(a):.4f} max {a.max():.4f}; top10 {np.round(flat[:10], 4)}")
    for nm in ("bf16", "fp16"):
        d = np.abs(outs[nm][li] - a)
        top = a > flat[min(2000, flat.size - 1)]
        print(f"   {nm}: max|err| all {d.max():.5f}  at the top-2000 logits {d[top].max():.5f}  mean {d.mean():.6f}")
allv = np.concatenate([o.ravel() for o in outs["fp32"]])
srt = np.sort(allv)[::-1]
for n in (50, 100, 200, 400, 800, 1600, 3200):
    gaps = srt[:n] - srt[1:n + 1]
    j = int(np.argmax(gaps[10:])) + 10
    print(f"top-{n}: widest gap {gaps[j]:.5f} after rank {j} (logit {srt[j]:.4f}); range of the top-{n}: {srt[0] - srt[n]:.4f}")
