"""GPU micro-benchmarks of the MFMA kernels on the real layer shapes (bs=12, 500x500): tile / split-K sweeps."""
import os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops, _hip

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us

N = 12
LAYERS = [  # name, H, W, Cin, Cout, K, stride
    ("l1.c1", 125, 125, 256, 64, 1, 1), ("l1.c2", 125, 125, 64, 64, 3, 1), ("l1.c3", 125, 125, 64, 256, 1, 1),
    ("l2.c1", 63, 63, 512, 128, 1, 1), ("l2.c2", 63, 63, 128, 128, 3, 1), ("l2.c3", 63, 63, 128, 512, 1, 1),
    ("l3.c1", 32, 32, 1024, 256, 1, 1), ("l3.c2", 32, 32, 256, 256, 3, 1), ("l3.c3", 32, 32, 256, 1024, 1, 1),
]
dt = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for name, H, W, Cin, Cout, K, s in LAYERS:
    p = K // 2
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, dt)
    wt = ops.pack_weight(w, dt, transpose=True)
    gy = torch.randn(N, H, W, Cout, device="cuda").to(dt)
    flops = 2.0 * N * H * W * Cin * Cout * K * K
    if which in ("conv", "all"):
        line = f"{name:6s} fwd  GF={flops/1e9:6.1f}"
        for tile in (13, 14, 12):
            us = timeit(lambda: ops.conv2d_nhwc(x, wp, Cout, K, K, s, p, tile=tile))
            line += f" | t{tile}: {us:7.1f}us {flops/us/1e6:6.1f}TF"
        print(line, flush=True)
        line = f"{name:6s} dgrd GF={flops/1e9:6.1f}"
        for tile in (13, 14, 12):
            us = timeit(lambda: ops.conv2d_nhwc(gy, wt, Cin, K, K, s, p, mode=1, out_hw=(H, W), tile=tile))
            line += f" | t{tile}: {us:7.1f}us {flops/us/1e6:6.1f}TF"
        print(line, flush=True)
    if which in ("wgrad", "all"):
        out = torch.zeros(Cout, Cin, K, K, device="cuda")
        for tile in (64, 1):
            line = f"{name:6s} wgrad tile{tile:3d}"
            for sk in (0, 4, 8, 16, 32):
                us = timeit(lambda: ops.conv2d_wgrad(x, gy, Cin, Cout, K, K, s, p, splitk=sk, tile=tile, out=out, packed=(K > 1)))
                line += f" | sk{sk}: {us:6.1f}us {flops/us/1e6:5.0f}TF"
            print(line, flush=True)
