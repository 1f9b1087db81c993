"""Do the pyramid levels of forward_levels really run side by side?  Host enqueue time of each level's forward and the device time line of the three
lanes (events recorded on each lane's stream before / after its forward), resident 480x640 + 960x1280 + 1920x2560 levels, bf16."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from bench import tame_init_
from tinyfaces.models.model import DetectionModel
dev = torch.device("cuda")
torch.manual_seed(0)
m = tame_init_(DetectionModel(num_templates=25)).to(dev).set_compute_dtype("bf16").eval()
xs = [torch.randn(1, 3, h, w, device=dev) for h, w in ((480, 640), (960, 1280), (1920, 2560))]
orig_f, orig_l = m._run_forward, m._run_eval_lane
log = []
def timed(fn, name):
    def w(x, *a, **k):
        t = time.perf_counter(); r = fn(x, *a, **k); log.append((name, tuple(x.shape[2:]), (time.perf_counter() - t) * 1e3)); return r
    return w
m._run_forward = timed(orig_f, "lane0"); m._run_eval_lane = timed(orig_l, "lane")
with torch.no_grad(), m.constant_weights(reserve=(1, 1920, 2560)):
    for rep in range(4):
        log.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = m.forward_levels(xs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"[{rep}] host enqueue {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms; per level (host ms): " + ", ".join(f"{n}{s}: {ms:.2f}" for n, s, ms in log), flush=True)
    # each level alone
    for x in xs:
        torch.cuda.synchronize(); t = time.perf_counter(); m(x); torch.cuda.synchronize()
        print(f"alone {tuple(x.shape[2:])}: {1e3 * (time.perf_counter() - t):.2f} ms")
    os.environ["TINYFACES_EVAL_LANES"] = "1"
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter(); m.forward_levels(xs); torch.cuda.synchronize()
        print(f"lanes=1 (sequential): {1e3 * (time.perf_counter() - t):.2f} ms")
