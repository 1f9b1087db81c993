"""NMS alone at the configs[4] sizes (N = 65 536 boxes of the template sizes on a 3750 x 5000 canvas; 8 lists of 8 192; the 12 k-candidate
list of the configs[1] pyramid) -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (rank / mask / scan)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
from tinyfaces.datasets.templates import load_templates
templates = load_templates()
rng = np.random.RandomState(3)
def boxes(n, w, h):
    t = templates[rng.randint(0, templates.shape[0], n)]
    bw, bh = t[:, 2] - t[:, 0] + 1, t[:, 3] - t[:, 1] + 1
    cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
    return np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1), rng.randn(n)
def t_of(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
for n, (w, h) in ((65536, (5000, 3750)), (12288, (2560, 1920)), (4096, (1280, 960))):
    b, s = boxes(n, w, h)
    B, S = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    ms, k = t_of(lambda: ops.nms(B, S, 0.3))
    print(f"N={n}: {ms:.3f} ms, kept {k.numel()}")
b8, s8 = boxes(65536, 2500, 1875)
B8, S8 = torch.from_numpy(b8).cuda(), torch.from_numpy(s8).cuda()
ms, k = t_of(lambda: ops.nms_batched(B8, S8, [8192 * i for i in range(9)], 0.3))
print(f"8 x 8192 batched: {ms:.3f} ms, kept {sum(x.numel() for x in k)}")
