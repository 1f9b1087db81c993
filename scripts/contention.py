"""Two-stream contention, measured: which CU resource do the data-gradient chain and the weight-gradient
stream fight over?  rocprofv3 --pmc serialises dispatches, so this is a wall-clock experiment: every VICTIM (a real layer-3
kernel of the bs=12 step, launched R times back to back on one stream) is timed alone and beside every AGGRESSOR running on a
second stream -- the real weight-gradient kernels, and the single-resource probes of csrc/probe.hip (park = LDS capacity / wave
slots only, L2 loads, HBM loads, MFMA, LDS-DMA, fp32 atomics, LDS reads).  Output: a table of microseconds per victim launch
and the slow-down factor; a run is only counted when the aggressor was still busy after the victim loop ended.

    python scripts/contention.py > gpurun_out/contention.txt
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops  # noqa: E402

dev = torch.device("cuda")
dt = torch.bfloat16
N, H, W = 12, 32, 32
M = N * H * W
R = int(os.environ.get("R", "24"))
lib = _hip.lib()
main, side = torch.cuda.Stream(), torch.cuda.Stream()


def rnd(*s):
    return torch.randn(*s, device=dev).to(dt)


# ---- operands of one layer-3 bottleneck (planes 256)
x1024, x256 = rnd(N, H, W, 1024), rnd(N, H, W, 256)
g1024, g256 = rnd(N, H, W, 1024), rnd(N, H, W, 256)
aux1024a, aux1024b, aux1024c = rnd(N, H, W, 1024), rnd(N, H, W, 1024), rnd(N, H, W, 1024)
w1 = torch.randn(256, 1024, 1, 1, device=dev) * 0.03
w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.03
w3 = torch.randn(1024, 256, 1, 1, device=dev) * 0.03
w1p, w1t = ops.pack_weight(w1, dt), ops.pack_weight(w1, dt, transpose=True)
w2p, w2t = ops.pack_weight(w2, dt), ops.pack_weight(w2, dt, transpose=True)
w3p, w3t = ops.pack_weight(w3, dt), ops.pack_weight(w3, dt, transpose=True)
sc256, sh256 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
dw1, dw2, dw3 = torch.zeros(256, 1024, 1, 1, device=dev), torch.zeros(256, 256, 3, 3, device=dev), torch.zeros(1024, 256, 1, 1, device=dev)
E = _hip


def conv(x, w_packed, Cout, K, mode=0, epi=0, aux=None, aux2=None, aux3=None, mask=None):
    """a prepared tf_conv2d launch (operands, output and statistic rows allocated once: the timed loop is launches only)"""
    a = _hip.ConvArgs()
    a.dtype, a.mode = _hip.TF_BF16, mode
    a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, x.shape[3], H, W, Cout, K, K, 1, K // 2
    a.ldy, a.epi, a.tile = Cout, epi, 0
    y = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
    a.x, a.w, a.y = x.data_ptr(), w_packed.data_ptr(), y.data_ptr()
    keep = [x, w_packed, y, aux, aux2, aux3, mask]
    for name, t in (("aux", aux), ("aux2", aux2), ("aux3", aux3)):
        if t is not None:
            setattr(a, name, t.data_ptr())
    if mask is not None:
        a.mask_scale, a.mask_shift = mask[0].data_ptr(), mask[1].data_ptr()
    if epi & (E.EPI_STATS | E.EPI_STATS2 | E.EPI_STATS3):
        st = torch.zeros(lib.tf_conv_mtiles(C.byref(a)), 2, Cout, device=dev)
        a.stat_out = st.data_ptr()
        keep.append(st)

    def fn():
        _hip.check(lib.tf_conv2d(C.byref(a), _hip.stream()), "tf_conv2d")
    fn.keep = keep
    return fn


def bn_apply(C_):
    g = g1024 if C_ == 1024 else g256
    x = x1024 if C_ == 1024 else x256
    out = torch.empty_like(g)
    rows = lib.tf_get_stat_rows()
    stat = torch.randn(rows, 2, C_, device=dev)
    gamma, mean, invstd = torch.ones(C_, device=dev), torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
    dgam, dbet = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    d = _hip.BnBwdDesc()
    d.stat, d.gamma, d.mean, d.invstd, d.dgamma, d.dbeta, d.nk, d.kidx = stat.data_ptr(), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), 2, 1
    keep = (stat, gamma, mean, invstd, dgam, dbet, out, d)

    def fn():
        _hip.check(lib.tf_bn_bwd_apply_fused(_hip.TF_BF16, g.data_ptr(), None, x.data_ptr(), C.byref(d), rows, M, C_, float(M), out.data_ptr(), _hip.stream()), "apply")
    fn.keep = keep
    return fn


VICTIMS = {
    "fwd  c1 1024>256 stats": conv(x1024, w1p, 256, 1, epi=E.EPI_STATS),
    "fwd  c2 3x3 256 stats ": conv(x256, w2p, 256, 3, epi=E.EPI_STATS),
    "fwd  c3 256>1024 stats": conv(x256, w3p, 1024, 1, epi=E.EPI_STATS),
    "dgrd c3 1024>256 m+s2 ": conv(g1024, w3t, 256, 1, mode=1, epi=E.EPI_MASK | E.EPI_STATS2, aux=x256, mask=(sc256, sh256)),
    "dgrd c2 3x3 256 m+s2  ": conv(g256, w2t, 256, 3, mode=1, epi=E.EPI_MASK | E.EPI_STATS2, aux=x256, mask=(sc256, sh256)),
    "dgrd c1 256>1024 hand ": conv(g256, w1t, 1024, 1, mode=1, epi=E.EPI_RES | E.EPI_MASK2 | E.EPI_STATS3, aux=aux1024a, aux2=aux1024b, aux3=aux1024c),
    "bn_bwd_apply C=1024   ": bn_apply(1024),
    "bn_bwd_apply C=256    ": bn_apply(256),
}
WG = {
    "wgrad c3 (dy1024,x256)": lambda: ops.conv2d_wgrad(x256, g1024, 256, 1024, 1, 1, 1, 0, out=dw3),
    "wgrad c1 (dy256,x1024)": lambda: ops.conv2d_wgrad(x1024, g256, 1024, 256, 1, 1, 1, 0, out=dw1),
    "wgrad c2 3x3 two-phase": lambda: ops.conv2d_wgrad(x256, g256, 256, 256, 3, 3, 1, 1, out=dw2, two_phase=True),
}
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)          # 1 GiB window for the HBM probe
small = torch.zeros(2 << 20, dtype=torch.uint8, device=dev)         # 2 MiB: L2-resident on every XCD


def probe(kind, blocks, lds=0, window=None, iters=1):
    buf = big if kind == 2 else small
    wb = buf.numel() if window is None else window
    return lambda it=iters: _hip.check(lib.tf_debug_probe(kind, blocks, lds, buf.data_ptr(), wb, it, _hip.stream()), "probe")


PROBES = {
    "park 1024blk lds0     ": probe(0, 1024),
    "park 256blk lds64K    ": probe(0, 256, 64 << 10),
    "park 256blk lds128K   ": probe(0, 256, 128 << 10),
    "park 512blk lds48K    ": probe(0, 512, 48 << 10),
    "L2 loads 512blk       ": probe(1, 512),
    "HBM loads 512blk      ": probe(2, 512),
    "MFMA 512blk           ": probe(3, 512),
    "MFMA 256blk           ": probe(3, 256),
    "LDS-DMA 512blk        ": probe(4, 512),
    "atomics 512blk        ": probe(5, 512),
    "LDS reads 512blk      ": probe(6, 512),
}


def time_loop(fn, reps, stream):
    with torch.cuda.stream(stream):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream.synchronize()
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
        b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def calib_probe(fn):
    """iterations for ~1 us of the probe alone"""
    with torch.cuda.stream(side):
        fn(64); side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side); fn(4096); b.record(side); b.synchronize()
    return 4096.0 / (a.elapsed_time(b) * 1e3)          # iterations per microsecond


def beside(victim, alone_us, start_aggr):
    """victim loop on `main` while the aggressor runs on `side`; returns (us per launch, aggressor still running at the end)"""
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        start_aggr(alone_us * R * 8.0)                    # enough aggressor work for an 8x slow-down
        done.record(side)
    with torch.cuda.stream(main):
        for _ in range(3):
            victim()                                       # let the aggressor ramp up
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        for _ in range(R):
            victim()
        b.record(main)
        b.synchronize()
    still = not done.query()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / R * 1e3, still


def main_():
    print(f"# contention micro-benchmark, layer-3 shapes of the bs=12 step (M = {M}), R = {R} victim launches per cell; device {torch.cuda.get_device_name(0)}")
    alone = {k: time_loop(f, R, main) for k, f in VICTIMS.items()}
    wg_alone = {k: time_loop(f, R, side) for k, f in WG.items()}
    ipus = {k: calib_probe(f) for k, f in PROBES.items()}
    print("\n## alone (us per launch, back-to-back launches on one stream)")
    for k, v in list(alone.items()) + list(wg_alone.items()):
        print(f"{k} {v:8.1f}")
    print("\n## probes alone: iterations per microsecond")
    for k, v in ipus.items():
        print(f"{k} {v:10.2f}")

    aggr = {}
    for k, f in WG.items():
        aggr[k] = (lambda f_, t_: (lambda budget_us: [f_() for _ in range(int(budget_us / t_) + 2)]))(f, wg_alone[k])
    for k, f in PROBES.items():
        aggr[k] = (lambda f_, r_: (lambda budget_us: f_(max(1, int(budget_us * r_)))))(f, ipus[k])

    print("\n## victim beside aggressor: us per victim launch (x slow-down); '!' = the aggressor ended before the victim loop did")
    names = list(aggr)
    print("victim".ljust(24) + " | alone | " + " | ".join(n.strip()[:18].ljust(18) for n in names))
    for vk, vf in VICTIMS.items():
        cells = []
        for an in names:
            us, still = beside(vf, alone[vk], aggr[an])
            cells.append(f"{us:6.1f} x{us / alone[vk]:4.2f}{' ' if still else '!'}".ljust(18))
        print(vk.ljust(24) + f" | {alone[vk]:5.1f} | " + " | ".join(cells), flush=True)

    print("\n## the reverse: weight gradient (victim, on the side stream's role) beside the chain kernels looping on the other stream")
    vnames = list(VICTIMS)
    print("victim".ljust(24) + " | alone | " + " | ".join(n.strip()[:18].ljust(18) for n in vnames))
    for wk, wf in WG.items():
        cells = []
        for vk in vnames:
            start = (lambda f_, t_: (lambda budget_us: [f_() for _ in range(int(budget_us / t_) + 2)]))(VICTIMS[vk], alone[vk])
            us, still = beside(wf, wg_alone[wk], start)
            cells.append(f"{us:6.1f} x{us / wg_alone[wk]:4.2f}{' ' if still else '!'}".ljust(18))
        print(wk.ljust(24) + f" | {wg_alone[wk]:5.1f} | " + " | ".join(cells), flush=True)

    # the pointwise weight gradient's block count (split-K): fewer blocks = less pressure on the chain, slower gradient
    print("\n## split-K of the pointwise weight gradients: alone, and the hand-over data gradient beside it")
    for name, mk in (("wgrad c3", lambda sk: (lambda: ops.conv2d_wgrad(x256, g1024, 256, 1024, 1, 1, 1, 0, out=dw3, splitk=sk))),
                     ("wgrad c1", lambda sk: (lambda: ops.conv2d_wgrad(x1024, g256, 1024, 256, 1, 1, 1, 0, out=dw1, splitk=sk)))):
        for sk in (1, 2, 4, 8, 16):
            f = mk(sk)
            t = time_loop(f, R, side)
            start = (lambda f_, t_: (lambda budget_us: [f_() for _ in range(int(budget_us / t_) + 2)]))(f, t)
            vk = "dgrd c1 256>1024 hand "
            us, still = beside(VICTIMS[vk], alone[vk], start)
            us2, still2 = beside(VICTIMS["dgrd c2 3x3 256 m+s2  "], alone["dgrd c2 3x3 256 m+s2  "], start)
            print(f"{name} splitk={sk:2d} ({64 * sk:4d} blocks): alone {t:6.1f} us | hand-over dgrad beside it {us:6.1f} (x{us / alone[vk]:4.2f}){'' if still else '!'}"
                  f" | 3x3 dgrad beside it {us2:6.1f} (x{us2 / alone['dgrd c2 3x3 256 m+s2  ']:4.2f}){'' if still2 else '!'}", flush=True)


main_()
