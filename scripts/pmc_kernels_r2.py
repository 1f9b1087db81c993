"""A handful of launches of the conv tile candidates and the 3x3 weight-gradient kernels on the layer-3 / layer-2 shapes (bs=12,
500x500) for rocprofv3 --pmc passes (`SCRIPT=scripts/pmc_kernels_r2.py scripts/gpu_job.sh pmc-all`)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
N, dt = 12, torch.bfloat16
for name, H, W, Cin, Cout, K in (("l3.c1", 32, 32, 1024, 256, 1), ("l3.c2", 32, 32, 256, 256, 3), ("l2.c2", 63, 63, 128, 128, 3)):
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, dt)
    gy = torch.randn(N, H, W, Cout, device="cuda").to(dt)
    out = torch.zeros(Cout, Cin, K, K, device="cuda")
    for tile in (13, 44, 46) + ((50,) if K == 3 else ()):       # 50: the halo-resident 3x3 kernel (conv3x3h)
        for _ in range(3):
            ops.conv2d_nhwc(x, wp, Cout, K, K, 1, K // 2, tile=tile)
    if K == 3:
        for _ in range(3):
            ops.conv2d_wgrad(x, gy, Cin, Cout, K, K, 1, 1, out=out, packed=True, tile=1)
            ops.conv2d_wgrad(x, gy, Cin, Cout, K, K, 1, 1, out=out, packed=True, tile=3, two_phase=True)
torch.cuda.synchronize()
