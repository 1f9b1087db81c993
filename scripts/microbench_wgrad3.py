"""3x3 weight gradients on the real layer shapes: the per-tap LDS-DMA kernel (tile 1) vs the all-taps kernel (tile 3), several
split-K factors.  Run under rocprofv3 --kernel-trace for true durations (`scripts/gpu_job.sh ubench scripts/microbench_wgrad3.py`)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

N = 12
for name, H, W, C in [("l1.c2", 125, 125, 64), ("l2.c2", 63, 63, 128), ("l3.c2", 32, 32, 256)]:
    x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
    gy = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
    out = torch.zeros(C, C, 3, 3, device="cuda")
    flops = 2.0 * N * H * W * C * C * 9
    ref = ops.conv2d_wgrad(x, gy, C, C, 3, 3, 1, 1, tile=1, packed=True).clone()
    line = f"{name} GF={flops/1e9:5.1f}"
    us = timeit(lambda: ops.conv2d_wgrad(x, gy, C, C, 3, 3, 1, 1, tile=1, out=out, packed=True))
    line += f" | per-tap: {us:6.1f}us {flops/us/1e6:4.0f}TF"
    tiles = (C // 64) ** 2
    for blocks in (128, 256, 512):
        sk = max(1, blocks // tiles)
        out.zero_()
        got = ops.conv2d_wgrad(x, gy, C, C, 3, 3, 1, 1, tile=3, splitk=sk, out=out, packed=True)
        dif = float((got - ref).abs().max() / ref.abs().max())
        us = timeit(lambda: ops.conv2d_wgrad(x, gy, C, C, 3, 3, 1, 1, tile=3, splitk=sk, out=out, packed=True))
        us2 = timeit(lambda: ops.conv2d_wgrad(x, gy, C, C, 3, 3, 1, 1, tile=3, splitk=sk, out=out, packed=True, two_phase=True))
        line += f" | all-taps sk{sk}: atomics {us:6.1f}us, two-phase {us2:6.1f}us {flops/us2/1e6:4.0f}TF d={dif:.0e}"
    print(line, flush=True)
