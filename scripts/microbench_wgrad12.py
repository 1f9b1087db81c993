"""Standalone timings (HIP events, back-to-back launches) of the layer-1 / layer-2 weight gradients as the step launches them (bs = 12, 500 x 500),
beside the HBM time of their operands read once (8 TB/s)."""
import os, sys
import torch
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
SHAPES = [  # name, launches per step, H, W, OH, OW, Cin, Cout, K, stride
    ("l1.c1 64->64", 1, 125, 125, 125, 125, 64, 64, 1, 1), ("l1.c1 256->64", 2, 125, 125, 125, 125, 256, 64, 1, 1),
    ("l1.c2 3x3 64", 3, 125, 125, 125, 125, 64, 64, 3, 1), ("l1.c3/ds 64->256", 4, 125, 125, 125, 125, 64, 256, 1, 1),
    ("l2.c1 256->128", 1, 125, 125, 125, 125, 256, 128, 1, 1), ("l2.c2 3x3 s2", 1, 125, 125, 63, 63, 128, 128, 3, 2),
    ("l2.ds 256->512 s2", 1, 125, 125, 63, 63, 256, 512, 1, 2), ("l2.c1 512->128", 3, 63, 63, 63, 63, 512, 128, 1, 1),
    ("l2.c2 3x3 128", 3, 63, 63, 63, 63, 128, 128, 3, 1), ("l2.c3 128->512", 4, 63, 63, 63, 63, 128, 512, 1, 1),
    ("l3.c1 512->256", 1, 63, 63, 63, 63, 512, 256, 1, 1), ("l3.c2 3x3 s2", 1, 63, 63, 32, 32, 256, 256, 3, 2),
    ("l3.ds 512->1024 s2", 1, 63, 63, 32, 32, 512, 1024, 1, 2), ("l3.c3 256->1024", 1, 32, 32, 32, 32, 256, 1024, 1, 1),
]
dt = torch.bfloat16
tot = 0.0
for name, n, H, W, OH, OW, Cin, Cout, K, s in SHAPES:
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    gy = torch.randn(N, OH, OW, Cout, device="cuda").to(dt)
    out = torch.zeros(Cout, Cin, K, K, device="cuda")
    two = K == 3 and s == 1
    us = timeit(lambda: ops.conv2d_wgrad(x, gy, Cin, Cout, K, K, s, K // 2, out=out, packed=False, two_phase=two))
    by = (x.numel() + gy.numel()) * 2 + out.numel() * 4
    fl = 2.0 * N * OH * OW * Cin * Cout * K * K
    tot += us * n
    print(f"{name:20s} x{n}  {us:7.1f} us   operands {by / 1e6:6.1f} MB = {by / 8e6:5.1f} us at 8 TB/s  ({us / (by / 8e6):4.1f}x)   {fl / us / 1e6:6.0f} TFLOP/s", flush=True)
print(f"sum over the step's launches: {tot:.0f} us")
