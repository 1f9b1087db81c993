"""Round-2 micro-benchmark of the conv tiles on the real layer shapes (bs=12, 500x500), meant to run under
`rocprofv3 --kernel-trace` (`scripts/gpu_job.sh ubench scripts/microbench_r2.py`): TRUE kernel durations come from the trace, the printed event
timings include the Python launch path.  Every variant is also checked against the 64x64 reference tile (max abs difference)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

N = 12
LAYERS = [("l1.c1", 125, 125, 256, 64, 1), ("l1.c2", 125, 125, 64, 64, 3), ("l1.c3", 125, 125, 64, 256, 1),
          ("l2.c1", 63, 63, 512, 128, 1), ("l2.c2", 63, 63, 128, 128, 3), ("l2.c3", 63, 63, 128, 512, 1),
          ("l3.c1", 32, 32, 1024, 256, 1), ("l3.c2", 32, 32, 256, 256, 3), ("l3.c3", 32, 32, 256, 1024, 1)]
TILES = [int(t) for t in os.environ.get("TILES", "0,13,14,44,15,45,16,46").split(",")]
REPS = int(os.environ.get("REPS", "12"))
only = os.environ.get("LAYERS", "")
dt = torch.bfloat16
for name, H, W, Cin, Cout, K in LAYERS:
    if only and name not in only.split(","):
        continue
    p = K // 2
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, K, K, device="cuda") * 0.05
    wp, wt = ops.pack_weight(w, dt), ops.pack_weight(w, dt, transpose=True)
    gy = torch.randn(N, H, W, Cout, device="cuda").to(dt)
    aux = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    flops = 2.0 * N * H * W * Cin * Cout * K * K
    ref_f = ops.conv2d_nhwc(x, wp, Cout, K, K, 1, p, tile=13).float()
    ref_d = ops.conv2d_nhwc(gy, wt, Cin, K, K, 1, p, mode=1, out_hw=(H, W), tile=13).float()
    for kind in ("fwd", "dgrd", "hand"):
        if kind == "hand" and K != 1:
            continue
        line = f"{name:6s} {kind:4s} GF={flops/1e9:5.1f}"
        for tile in TILES:
            if kind == "fwd":
                fn = lambda: ops.conv2d_nhwc(x, wp, Cout, K, K, 1, p, epi=_hip.EPI_STATS, want_stats=True, tile=tile)
                dif = float((fn()[0].float() - ref_f).abs().max())
            elif kind == "dgrd":
                fn = lambda: ops.conv2d_nhwc(gy, wt, Cin, K, K, 1, p, mode=1, out_hw=(H, W), tile=tile)
                dif = float((fn().float() - ref_d).abs().max())
            else:       # the hand-over epilogue: RES | MASK2 | STATS3
                fn = lambda: ops.conv2d_nhwc(gy, wt, Cin, K, K, 1, p, mode=1, out_hw=(H, W), tile=tile, epi=_hip.EPI_RES | _hip.EPI_MASK2 | _hip.EPI_STATS3,
                                             aux=aux, aux2=aux, aux3=aux, want_stats=True)
                dif = 0.0
            us = timeit(fn, REPS)
            line += f" | t{tile}: {us:6.1f}us {flops/us/1e6:5.0f}TF d={dif:.0e}"
        print(line, flush=True)
