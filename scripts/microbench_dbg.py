"""conv_dma pipeline ablation on layer3 shapes: TF_CONV_DBG=0 normal, 1 no steady-state DMA, 2 no LDS reads/MFMA, 3 neither."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
N = 12
for name, H, W, Cin, Cout, K in [("l3.c1", 32, 32, 1024, 256, 1), ("l3.c2", 32, 32, 256, 256, 3), ("l3.c3", 32, 32, 256, 1024, 1), ("l1.c3", 125, 125, 64, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, torch.bfloat16)
    line = f"{name} dbg={os.environ.get('TF_CONV_DBG','0')}"
    for tile in [int(t) for t in os.environ.get("TILES", "13,11,12").split(",")]:
        us = timeit(lambda: ops.conv2d_nhwc(x, wp, Cout, K, K, 1, K // 2, tile=tile))
        line += f" | t{tile}: {us:6.1f}us"
    print(line, flush=True)
