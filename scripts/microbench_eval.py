"""Tile sweep of the pointwise convs of the evaluation graph (BN folded: affine [+ residual] + ReLU epilogue) on the layer shapes of the
1920x2560 pyramid level (configs[1]) -- the launches that dominate the per-image time once the small levels hide in the lanes."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops
from tinyfaces._hip import lib, ptr, stream
dt = torch.bfloat16
R = 20
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / R * 1e3
TILES = [int(t) for t in os.environ.get("TILES", "0,32,13,12,46,45,60").split(",")]
SHAPES = [("l3.conv2", 120, 160, 256, 256, False, 3), ("l2.conv2", 240, 320, 128, 128, False, 3), ("l1.conv2", 480, 640, 64, 64, False, 3),
          ("l3.conv3", 120, 160, 256, 1024, True), ("l3.conv1", 120, 160, 1024, 256, False), ("l2.conv3", 240, 320, 128, 512, True),
          ("l2.conv1", 240, 320, 512, 128, False), ("l1.conv3", 480, 640, 64, 256, True), ("l1.conv1", 480, 640, 256, 64, False)]
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if s[0] in os.environ["ONLY"].split(",")]
for name, H, W, K, Co, res, *ks in SHAPES:
    ks = ks[0] if ks else 1
    M = H * W
    x = torch.randn(1, H, W, K, device="cuda").to(dt)
    w = ops.pack_weight(torch.randn(Co, K, ks, ks, device="cuda") / (K * ks * ks) ** 0.5, dt)
    r = torch.randn(1, H, W, Co, device="cuda").to(dt)
    y = torch.empty(1, H, W, Co, dtype=dt, device="cuda")
    sc, sh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.1
    nbytes = (M * K + M * Co * (2 if res else 1) + K * Co) * 2
    line = f"{name} M={M} K={K * ks * ks} N={Co} hbm-bound {nbytes / 6.3e6:6.1f} us mfma-bound {2.0 * M * K * ks * ks * Co / 2.5e9:6.1f} us |"
    ref = None
    for t in TILES:
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 0
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = 1, H, W, K, H, W, Co, ks, ks, 1, ks // 2
        a.ldy, a.tile = Co, t
        a.epi = _hip.EPI_AFFINE | _hip.EPI_RELU | (_hip.EPI_RES if res else 0)
        a.x, a.w, a.y, a.aux, a.epi_scale, a.epi_shift = ptr(x), ptr(w), ptr(y), ptr(r), ptr(sc), ptr(sh)
        rc = lib().tf_conv2d(C.byref(a), stream())
        if rc != 0:
            line += f" t{t}: n/a |"
            continue
        us = timeit(lambda: lib().tf_conv2d(C.byref(a), stream()))
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        ok = bool(torch.allclose(y.float(), ref.float(), rtol=2e-2, atol=2e-2))
        line += f" t{t}: {us:6.1f}{'' if ok else '!'} |"
    print(line, flush=True)
