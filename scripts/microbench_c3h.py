"""conv3x3h (tile 50) alone: input-channel sweep at the layer-3 training shape and at the large pyramid level.
Slope over the stage count (9 stages per 64 input channels) = cycles per K stage; intercept = prologue + epilogue.
Run under the kernel tracer (scripts/gpu_job.sh ubench) for true durations; the event timings printed here include launch gaps."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops, _hip

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

dt = torch.bfloat16
TILE = int(os.environ.get("C3H_TILE", "50"))
shapes = [(12, 32, 32), (1, 120, 160), (1, 60, 80), (12, 63, 63)]
CINS = [int(v) for v in os.environ.get("C3H_CINS", "64,128,256,512,1024").split(",")]
for (N, H, W) in shapes:
    for Cin in CINS:
        for Cout in (256,):
            x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
            w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
            wp = ops.pack_weight(w, dt)
            try:
                us = timeit(lambda: ops.conv2d_nhwc(x, wp, Cout, 3, 3, 1, 1, tile=TILE))
            except Exception as e:
                print(f"N{N} {H}x{W} Cin {Cin}: {e}"); continue
            fl = 2.0 * N * H * W * Cin * Cout * 9
            print(f"shape {N}x{H}x{W} Cin {Cin:5d} Cout {Cout}: stages {9 * Cin // 64:4d}  {us:7.1f} us  {fl / us / 1e6:6.1f} TF", flush=True)
