"""tf_upsample_add_crop (the last kernel of every evaluation forward: NHWC 2-byte score maps -> NCHW fp32 output) at the three pyramid level sizes and
at the training size: time, bytes, GB/s."""
import os, sys, torch
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
C, ldc = 125, 128
diag = torch.rand(C * 16, device="cuda")
for B, H3, W3 in ((1, 60, 80), (1, 120, 160), (1, 240, 320), (12, 63, 63), (1, 469, 625)):
    H4, W4 = (H3 + 1) // 2, (W3 + 1) // 2
    for dt in (torch.bfloat16, torch.float16):
        s3 = torch.randn(B, H3, W3, ldc, device="cuda").to(dt); s4 = torch.randn(B, H4, W4, ldc, device="cuda").to(dt)
        out = torch.empty(B, C, H3, W3, device="cuda")
        tfd = _hip.tf_dtype(dt)
        us = timeit(lambda: lib().tf_upsample_add_crop(tfd, ptr(s3), ptr(s4), ptr(diag), B, C, ldc, H3, W3, H4, W4, ptr(out), stream()))
        by = s3.numel() * 2 + s4.numel() * 2 + out.numel() * 4
        print(f"B={B} {H3}x{W3} {str(dt)[6:]:9s} {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)
print("backward (tf_upsample_add_crop_bwd): g NCHW fp32 -> g3, g4 NHWC")
for B, H3, W3 in ((12, 63, 63), (2, 120, 160)):
    H4, W4 = (H3 + 1) // 2, (W3 + 1) // 2
    for dt in (torch.bfloat16, torch.float32):
        g = torch.randn(B, C, H3, W3, device="cuda")
        g3 = torch.empty(B, H3, W3, ldc, device="cuda", dtype=dt); g4 = torch.empty(B, H4, W4, ldc, device="cuda", dtype=dt)
        tfd = _hip.tf_dtype(dt)
        us = timeit(lambda: lib().tf_upsample_add_crop_bwd(tfd, ptr(g), ptr(diag), B, C, ldc, H3, W3, H4, W4, ptr(g3), ptr(g4), stream()))
        by = g.numel() * 4 + (g3.numel() + g4.numel()) * g3.element_size()
        print(f"B={B} {H3}x{W3} {str(dt)[6:]:9s} {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)
