"""conv_pwx (r3) against what it replaces, on the layer-3 / layer-2 shapes of the bs=12 step: (a) tf_bn_bwd_apply_fused + the pointwise data
gradient (conv_dma) vs tf_conv2d_bnbwd, (b) the plain pointwise conv (tile 13 vs tile 60).  HIP-event time of back-to-back launches;
run under rocprofv3 --kernel-trace --stats for the kernels' own durations."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops
from tinyfaces._hip import lib, ptr, stream
dt = torch.bfloat16
R = 30
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / R * 1e3
for name, (N, H, W, K, Co) in (("layer3", (12, 32, 32, 1024, 256)), ("layer2", (12, 63, 63, 512, 128))):
    M = N * H * W
    gz, x2 = torch.randn(N, H, W, K, device="cuda").to(dt), torch.randn(N, H, W, K, device="cuda").to(dt)
    w = torch.randn(K, Co, 1, 1, device="cuda") / K ** 0.5
    wt = ops.pack_weight(w, dt, transpose=True)
    cprev = torch.randn(N, H, W, Co, device="cuda").to(dt)
    ms, mh = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda") * 0.2
    rows = lib().tf_get_stat_rows()
    stat = torch.randn(rows, 2, K, device="cuda")
    gamma, mean, invstd = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.3, torch.rand(K, device="cuda") + 0.5
    dg, db = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    d = _hip.BnBwdDesc()
    d.stat, d.gamma, d.mean, d.invstd, d.dgamma, d.dbeta, d.nk, d.kidx = ptr(stat), ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db), 2, 1
    t1, y = torch.empty_like(gz), torch.empty(N, H, W, Co, dtype=dt, device="cuda")
    def args(x, tile=0):
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 1
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, K, H, W, Co, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Co, _hip.EPI_MASK | _hip.EPI_STATS2, tile
        st = torch.zeros(16, 2, Co, device="cuda")
        a.x, a.w, a.y, a.aux, a.mask_scale, a.mask_shift, a.stat_out = ptr(x), ptr(wt), ptr(y), ptr(cprev), ptr(ms), ptr(mh), ptr(st)
        a._keep = st
        return a
    a0, a1, a2 = args(t1), args(gz), args(gz, 60)
    def pair():
        lib().tf_bn_bwd_apply_fused(_hip.TF_BF16, ptr(gz), None, ptr(x2), C.byref(d), rows, M, K, float(M), ptr(t1), stream())
        lib().tf_conv2d(C.byref(a0), stream())
    fused = lambda: lib().tf_conv2d_bnbwd(C.byref(a1), C.byref(d), ptr(x2), ptr(t1), rows, float(M), stream())
    plain13 = lambda: lib().tf_conv2d(C.byref(a0), stream())
    plain60 = lambda: lib().tf_conv2d(C.byref(a2), stream())
    print(f"{name} M={M} K={K} N={Co}: apply+dgrad {timeit(pair):6.1f} us | conv_pwx fused {timeit(fused):6.1f} us | dgrad alone conv_dma {timeit(plain13):6.1f} | conv_pwx plain {timeit(plain60):6.1f}", flush=True)
    # r5: the forward pair  tf_bn_add_relu_fused + conv1 (conv_dma)  against tf_conv2d_bnfwd (same shapes: K = 4 planes -> planes)
    fstat = torch.randn(rows * 2 + 1, K, device="cuda").abs() * 100
    vec = [torch.zeros(K, device="cuda") for _ in range(4)] + [torch.zeros(K, device="cuda"), torch.ones(K, device="cuda")]
    fd = _hip.BnFwdDesc()
    fd.stat, fd.gamma, fd.beta = ptr(fstat), ptr(gamma), ptr(mean)
    fd.scale, fd.shift, fd.mean, fd.invstd, fd.running_mean, fd.running_var = [ptr(v) for v in vec]
    wf = ops.pack_weight(torch.randn(Co, K, 1, 1, device="cuda") / K ** 0.5, dt)
    yact = torch.empty_like(gz)
    def fargs(x, tile=0):
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 0
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, K, H, W, Co, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Co, _hip.EPI_STATS, tile
        st = torch.zeros(16, 2, Co, device="cuda")
        a.x, a.w, a.y, a.stat_out = ptr(x), ptr(wf), ptr(y), ptr(st)
        a._keep = st
        return a
    f0, f1 = fargs(yact), fargs(gz)
    def fpair():
        lib().tf_bn_add_relu_fused(_hip.TF_BF16, ptr(gz), C.byref(fd), ptr(x2), None, rows, M, K, float(M), 1e-5, 0.1, ptr(yact), stream())
        lib().tf_conv2d(C.byref(f0), stream())
    ffused = lambda: lib().tf_conv2d_bnfwd(C.byref(f1), C.byref(fd), ptr(x2), None, ptr(yact), rows, float(M), 1e-5, 0.1, stream())
    print(f"{name} forward: bn_add_relu+conv1 {timeit(fpair):6.1f} us | tf_conv2d_bnfwd {timeit(ffused):6.1f} us | conv1 alone {timeit(lambda: lib().tf_conv2d(C.byref(f0), stream())):6.1f}", flush=True)
