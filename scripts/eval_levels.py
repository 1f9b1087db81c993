"""r5: what does each level of the configs[1] pyramid cost?  forward_levels of resident level tensors inside one constant-weights session (what bench_eval
times, minus decode + NMS): every level alone, pairs, all three on 3 lanes and on 1."""
import os, sys, time, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd"]
import bench
from tinyfaces.models.model import DetectionModel
torch.manual_seed(0)
m = bench.tame_init_(DetectionModel(num_objects=1, num_templates=25)).set_compute_dtype("bf16").cuda().eval()
g = torch.Generator().manual_seed(0)
lv = {"small 480x640": torch.randn(1, 3, 480, 640, generator=g).cuda(), "mid 960x1280": torch.randn(1, 3, 960, 1280, generator=g).cuda(), "big 1920x2560": torch.randn(1, 3, 1920, 2560, generator=g).cuda()}
def t(xs, lanes, n=15):
    ts = []
    for i in range(n + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = m.forward_levels(xs, lanes=lanes) if len(xs) > 1 else [m(xs[0])]
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[3:])) * 1e3
with torch.no_grad(), m.constant_weights(reserve=(1, 1920, 2560)):
    for k, x in lv.items():
        print(f"{k:16s} alone          {t([x], 1):6.3f} ms", flush=True)
    xs = list(lv.values())
    print(f"big + mid        lanes 2        {t([xs[1], xs[2]], 2):6.3f} ms", flush=True)
    print(f"big + small      lanes 2        {t([xs[0], xs[2]], 2):6.3f} ms", flush=True)
    for lanes in (3, 2, 1):
        print(f"all three        lanes {lanes}        {t(xs, lanes):6.3f} ms", flush=True)
