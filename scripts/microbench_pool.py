"""tf_maxpool_fwd / tf_maxpool_bwd_stats at the training size (12 x 250 x 250 x 64 -> 125 x 125) and at the 1920 x 2560 pyramid level: us, GB/s."""
import os, sys, ctypes as C_, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
from tinyfaces._hip import lib
from tinyfaces.ops import ptr, stream
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
C = 64
for N, H, W in ((12, 250, 250), (1, 960, 1280)):
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dt = torch.bfloat16
    x = torch.randn(N, H, W, C, device="cuda").to(dt)
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    y = torch.empty(N, OH, OW, C, device="cuda", dtype=dt); idx = torch.empty(N, OH, OW, C, device="cuda", dtype=torch.uint8)
    tfd = _hip.tf_dtype(dt)
    us = timeit(lambda: lib().tf_maxpool_fwd(tfd, ptr(x), N, H, W, C, ptr(sc), ptr(sh), ptr(y), ptr(idx), stream()))
    by = x.numel() * 2 + y.numel() * 2 + idx.numel()
    print(f"fwd {N}x{H}x{W}: {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)
    g = torch.randn(N, OH, OW, C, device="cuda").to(dt); gz = torch.empty_like(x)
    rows = C_.c_int(0); st = torch.zeros(16, 2, C, device="cuda")
    us = timeit(lambda: lib().tf_maxpool_bwd_stats(tfd, ptr(g), ptr(idx), ptr(x), ptr(sc), ptr(sh), N, H, W, C, ptr(gz), ptr(st), C_.byref(rows), stream()))
    by = g.numel() * 2 + idx.numel() + x.numel() * 2 + gz.numel() * 2
    print(f"bwd+stats {N}x{H}x{W}: {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)
