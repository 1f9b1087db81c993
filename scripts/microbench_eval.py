"""Eval-shape conv micro-benchmark (batch 1, 1920x2560 pyramid level): tile candidates with the eval epilogues.
Run under rocprofv3 (scripts/gpu_ubench_trace.sh) for true kernel durations."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops, _hip
dt = torch.bfloat16
tiles = [int(t) for t in os.environ.get("TILES", "13,12,11").split(",")]
scale = float(os.environ.get("SCALE", "1.0"))
H1, W1 = int(480 * scale), int(640 * scale)
LAYERS = [("l1.c1", H1, W1, 256, 64, 1), ("l1.c2", H1, W1, 64, 64, 3), ("l1.c3", H1, W1, 64, 256, 1),
          ("l2.c1", H1 // 2, W1 // 2, 512, 128, 1), ("l2.c2", H1 // 2, W1 // 2, 128, 128, 3), ("l2.c3", H1 // 2, W1 // 2, 128, 512, 1),
          ("l3.c1", H1 // 4, W1 // 4, 1024, 256, 1), ("l3.c2", H1 // 4, W1 // 4, 256, 256, 3), ("l3.c3", H1 // 4, W1 // 4, 256, 1024, 1)]
for name, H, W, Cin, Cout, K in LAYERS:
    x = torch.randn(1, H, W, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, dt)
    sc, sh = torch.rand(Cout, device="cuda"), torch.rand(Cout, device="cuda")
    res = torch.randn(1, H, W, Cout, device="cuda").to(dt) if name.endswith("c3") else None
    epi = _hip.EPI_AFFINE | _hip.EPI_RELU | (_hip.EPI_RES if res is not None else 0)
    for tile in tiles:
        for _ in range(6):
            ops.conv2d_nhwc(x, wp, Cout, K, K, 1, K // 2, epi=epi, epi_scale=sc, epi_shift=sh, aux=res, tile=tile)
    torch.cuda.synchronize()
    print(name, H, W, Cin, Cout, K, "GF", 2.0 * H * W * Cin * Cout * K * K / 1e9, flush=True)
