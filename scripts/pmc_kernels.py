"""A handful of launches of the dominant kernels on the layer-3 shapes (bs=12, 500x500) for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
N, H, W, dt = 12, 32, 32, torch.bfloat16
for name, Cin, Cout, K in (("l3.c1", 1024, 256, 1), ("l3.c2", 256, 256, 3), ("l3.c3", 256, 1024, 1)):
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, dt)
    gy = torch.randn(N, H, W, Cout, device="cuda").to(dt)
    out = torch.zeros(Cout, Cin, K, K, device="cuda")
    for _ in range(3):
        ops.conv2d_nhwc(x, wp, Cout, K, K, 1, K // 2, tile=13)
        ops.conv2d_wgrad(x, gy, Cin, Cout, K, K, 1, K // 2, out=out, packed=(K > 1))
torch.cuda.synchronize()
