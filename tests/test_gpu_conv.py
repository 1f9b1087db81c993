"""-m gpu: the MFMA implicit-GEMM conv engine, the pixel-reduction wgrad and the HBM-bound
companions, through the C ABI, vs torch-CPU fp32 references of the same op.
Tolerances: fp32 path 1e-4 relative to the tensor's max (exact-fp32 MFMA, different summation
order); bf16 path compared with a reference fed the SAME bf16-rounded operands, so what is left
is fp32 accumulation order + one bf16 rounding of the output (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import err, from_nhwc, q, report, to_nhwc

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3}


def _g(seed):
    return torch.Generator().manual_seed(seed)


CONV_CASES = [
    # N, H, W, Cin, Cout, K, stride, pad
    (2, 17, 19, 64, 64, 1, 1, 0),
    (1, 33, 31, 128, 256, 1, 1, 0),
    (2, 20, 21, 64, 64, 3, 1, 1),
    (1, 21, 23, 128, 128, 3, 2, 1),
    (2, 15, 18, 256, 512, 1, 2, 0),
    (1, 9, 40, 512, 128, 1, 1, 0),        # Cout 125-style padding handled by ldy; here exact
    (3, 63, 63, 256, 64, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [0, 11, 12, 13, 22, 23, 32])
def test_conv_forward_plain(dtype, case, tile):
    from tinyfaces import ops
    N, H, W, Cin, Cout, K, s, p = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    ref = F.conv2d(q(x, dtype), q(w, dtype), stride=s, padding=p)
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, K, K, s, p, tile=tile)
    d = err(from_nhwc(y)[:, :Cout], ref)
    report(f"conv_fwd[{dtype},{case},t{tile}]", maxabs=d[0], rel=d[2])
    assert d[2] < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [12, 0, 13])
def test_conv_forward_fused_eval_epilogue(dtype, tile):
    """AFFINE (folded BN) + residual + ReLU, Cout=125 padded to 128 (the head shape)."""
    from tinyfaces import _hip, ops
    g = _g(3)
    N, H, W, Cin, Cout = 2, 13, 17, 512, 125
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(N, 128, H, W, generator=g)
    ref = torch.relu(F.conv2d(q(x, dtype), q(w, dtype)) * sc[:Cout].view(1, -1, 1, 1) + sh[:Cout].view(1, -1, 1, 1) + q(res, dtype)[:, :Cout])
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, 1, 1, 1, 0, ldy=128,
                        epi=_hip.EPI_AFFINE | _hip.EPI_RES | _hip.EPI_RELU, epi_scale=sc.cuda(), epi_shift=sh.cuda(), aux=to_nhwc(res, dtype), tile=tile)
    d = err(from_nhwc(y)[:, :Cout], ref)
    report(f"conv_epilogue_eval[{dtype}]", maxabs=d[0], rel=d[2])
    assert d[2] < TOL[dtype]


def test_conv_refuses_the_removed_register_staged_kernel():
    """r4: the register-staged conv kernel of round 1 (tile codes 1-3, the producer-BN prologue `pro`) is gone -- the executor has
    materialised relu(bn(x)) since round 1 -- and tf_conv2d says so loudly instead of ignoring the request."""
    from tinyfaces import ops
    x = torch.randn(1, 8, 8, 64, device="cuda").to(torch.bfloat16)
    w = ops.pack_weight(torch.randn(64, 64, 1, 1, device="cuda"), torch.bfloat16)
    ps, ph = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, w, 64, 1, 1, 1, 0, pro=(ps, ph, True))
    for tile in (1, 2, 3):
        with pytest.raises(RuntimeError):
            ops.conv2d_nhwc(x, w, 64, 1, 1, 1, 0, tile=tile)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 11, 12, 13, 32])
def test_conv_forward_stats_no_prologue(dtype, tile):
    """raw output + (sum, sumsq) partials without a prologue -> the LDS-DMA kernel's column-sum epilogue."""
    from tinyfaces import _hip, ops
    g = _g(41)
    N, H, W, Cin, Cout = 2, 19, 23, 256, 128
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    ref = F.conv2d(q(x, dtype), q(w, dtype), padding=1)
    y, st = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, 3, 3, 1, 1, epi=_hip.EPI_STATS, want_stats=True, tile=tile)
    d = err(from_nhwc(y), ref)
    ssum = st.sum(0).cpu()
    n = ref.numel() / Cout
    d1 = err(ssum[0] / n, ref.mean(dim=(0, 2, 3)))
    d2 = err(ssum[1] / n, (ref ** 2).mean(dim=(0, 2, 3)))
    report(f"conv_stats_dma[{dtype},t{tile}]", rel=d[2], mean_abs=d1[0], sq_rel=d2[2])
    assert d[2] < TOL[dtype] and d1[0] < 2e-3 and d2[2] < 2e-3


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 12, 13])
@pytest.mark.parametrize("K,s,H", [(1, 1, 14), (3, 1, 14), (3, 2, 14), (3, 2, 15), (1, 2, 15)])
def test_conv_dgrad_mode(dtype, K, s, H, tile):
    """mode 1 == data gradient of conv(stride, pad): compared with torch autograd."""
    from tinyfaces import ops
    g = _g(20 + K + s + H)
    N, W, Cin, Cout = 2, H + 3, 64, 128
    p = K // 2
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    yref = F.conv2d(x, q(w, dtype), stride=s, padding=p)
    gy = torch.randn(yref.shape, generator=g)
    yref.backward(q(gy, dtype))
    wt = ops.pack_weight(w.cuda(), dtype, transpose=True)
    gx = ops.conv2d_nhwc(to_nhwc(gy, dtype), wt, Cin, K, K, s, p, mode=1, out_hw=(H, W), tile=tile)
    d = err(from_nhwc(gx), x.grad)
    report(f"conv_dgrad[{dtype},k{K}s{s}h{H}]", maxabs=d[0], rel=d[2])
    assert d[2] < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [12, 0, 11])
def test_conv_dgrad_mask_stats2_and_join(dtype, tile):
    from tinyfaces import _hip, ops
    g = _g(33)
    N, H, W, C1, C2 = 2, 16, 18, 256, 64          # dgrad of a 1x1 conv C2 -> C1 : input grad has C2 channels
    gy = torch.randn(N, C1, H, W, generator=g)
    w = torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5
    craw = torch.randn(N, C2, H, W, generator=g)
    ms, mh = torch.rand(C2, generator=g) + 0.5, torch.randn(C2, generator=g) * 0.5
    base = F.conv_transpose2d(q(gy, dtype), q(w, dtype))
    mask = (q(craw, dtype) * ms.view(1, -1, 1, 1) + mh.view(1, -1, 1, 1)) > 0
    ref = base * mask
    y, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), ops.pack_weight(w.cuda(), dtype, transpose=True), C2, 1, 1, 1, 0, mode=1, out_hw=(H, W),
                            epi=_hip.EPI_MASK | _hip.EPI_STATS2, aux=to_nhwc(craw, dtype), mask=(ms.cuda(), mh.cuda()), want_stats=True, tile=tile)
    d = err(from_nhwc(y), ref)
    s = st.sum(0).cpu()
    d1 = err(s[0], ref.sum(dim=(0, 2, 3)))
    d2 = err(s[1], (ref * q(craw, dtype)).sum(dim=(0, 2, 3)))
    # JOIN: out = acc + (y2 > 0 ? g3 : 0)
    y2 = torch.randn(N, C2, H, W, generator=g)
    g3 = torch.randn(N, C2, H, W, generator=g)
    refj = base + q(g3, dtype) * (q(y2, dtype) > 0)
    yj = ops.conv2d_nhwc(to_nhwc(gy, dtype), ops.pack_weight(w.cuda(), dtype, transpose=True), C2, 1, 1, 1, 0, mode=1, out_hw=(H, W),
                         epi=_hip.EPI_JOIN, aux2=to_nhwc(y2, dtype), aux3=to_nhwc(g3, dtype), tile=tile)
    dj = err(from_nhwc(yj), refj)
    report(f"conv_dgrad_mask_join[{dtype}]", rel=d[2], s1=d1[2], s2=d2[2], join_rel=dj[2])
    assert d[2] < TOL[dtype] and dj[2] < TOL[dtype] and d1[2] < 5e-3 and d2[2] < 5e-3


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("tile", [0, 11, 12])
def test_conv_dgrad_res_mask2_stats3(dtype, tile):
    """The hand-over epilogue of a Bottleneck's last data-gradient conv: out = (conv + res) * (y_prev > 0), with the
    per-channel sums  sum(out), sum(out * c3_prev)  of the previous block's BN3 backward (LDS-DMA kernel only)."""
    from tinyfaces import _hip, ops
    g = _g(77)
    N, H, W, C1, C2 = 2, 17, 15, 64, 256          # dgrad of a 1x1 conv C2 -> C1: the input gradient has C2 channels
    gy = torch.randn(N, C1, H, W, generator=g)
    w = torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5
    res = torch.randn(N, C2, H, W, generator=g)
    yprev = torch.randn(N, C2, H, W, generator=g)
    c3 = torch.randn(N, C2, H, W, generator=g) * 1.5 + 0.3
    base = F.conv_transpose2d(q(gy, dtype), q(w, dtype)) + q(res, dtype)
    ref = base * (q(yprev, dtype) > 0)
    y, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), ops.pack_weight(w.cuda(), dtype, transpose=True), C2, 1, 1, 1, 0, mode=1, out_hw=(H, W),
                            epi=_hip.EPI_RES | _hip.EPI_MASK2 | _hip.EPI_STATS3, aux=to_nhwc(res, dtype), aux2=to_nhwc(yprev, dtype),
                            aux3=to_nhwc(c3, dtype), want_stats=True, tile=tile)
    d = err(from_nhwc(y), ref)
    s = st.sum(0).cpu()
    d1 = err(s[0], ref.sum(dim=(0, 2, 3)))
    d2 = err(s[1], (ref * q(c3, dtype)).sum(dim=(0, 2, 3)))
    report(f"conv_dgrad_handover[{dtype},t{tile}]", rel=d[2], s1=d1[2], s2=d2[2])
    assert d[2] < TOL[dtype] and d1[2] < 5e-3 and d2[2] < 5e-3


@pytest.mark.parametrize("dtype", DT)
def test_pack_weights_tiled_equals_elementwise_pack(dtype):
    """tf_pack_weights_tiled (one read of the fp32 master, both operand layouts, LDS-tiled) must write exactly what the
    per-element tf_pack_weight writes, padding included: 1x1 / 3x3, Cout not a multiple of the tile (heads: 125), the
    flattened stem weight (147 -> 192 columns), forward-only and transposed-only jobs, >40 jobs (two launches)."""
    import ctypes as C
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    g = _g(5)
    tfd = tf_dtype(dtype)
    shapes = [(64, 147, 1, 192, False), (64, 64, 1, 0, True), (64, 64, 3, 0, True), (256, 64, 1, 0, True), (128, 128, 3, 0, True),
              (125, 512, 1, 0, True), (125, 1024, 1, 0, True), (1024, 256, 1, 0, True), (256, 256, 3, 0, True)]
    shapes = shapes + [(64 + 8 * i, 64, 1, 0, True) for i in range(36)]           # > 40 jobs
    ws, jobs, outs = [], (_hip.Pack2Job * len(shapes))(), []
    for n, (co, ci, k, cpad, both) in enumerate(shapes):
        w = torch.randn(co, ci, k, k, generator=g).cuda()
        rp, cp = (co + 127) // 128 * 128, cpad or ci
        rpt, cpt = (ci + 127) // 128 * 128, (co + 3) // 4 * 4          # 4 elements per store: padded like the executor's head operands
        o = torch.full((rp, k * k, cp), 7.0, dtype=dtype, device="cuda")
        ot = torch.full((rpt, k * k, cpt), 7.0, dtype=dtype, device="cuda") if both else None
        if n == 1:
            o, ot = None, ot                                                       # transposed-only job
        jobs[n] = _hip.Pack2Job(ptr(w), ptr(o), ptr(ot), co, ci, k * k, rp, cp, rpt, cpt)
        ws.append(w); outs.append((o, ot))
    assert lib().tf_pack_weights_tiled(tfd, C.cast(jobs, C.c_void_p), len(shapes), stream()) == 0
    torch.cuda.synchronize()
    for (co, ci, k, cpad, both), w, (o, ot) in zip(shapes, ws, outs):
        if o is not None:
            ref = ops.pack_weight(w, dtype, cols_pad=cpad or None)
            assert torch.equal(o.view(-1), ref.view(-1)), (co, ci, k)
        if ot is not None:
            reft = ops.pack_weight(w, dtype, transpose=True, cols_pad=(co + 3) // 4 * 4)
            assert torch.equal(ot.view(-1), reft.view(-1)), (co, ci, k, "t")


REAL_CASES = [
    # the layer shapes of the bs=12 500x500 training step that bench.py times (N, H, W, Cin, Cout, K, stride, pad)
    (12, 125, 125, 64, 256, 1, 1, 0),      # layer1 conv3 / downsample: M = 187 500, K = 64 (one K-stage, 5 860-block grids)
    (12, 125, 125, 256, 64, 1, 1, 0),      # layer1 conv1
    (12, 125, 125, 64, 64, 3, 1, 1),       # layer1 conv2
    (12, 125, 125, 128, 128, 3, 2, 1),     # layer2.0 conv2 (stride 2): M = 47 628
    (12, 63, 63, 128, 512, 1, 1, 0),       # layer2 conv3
    (12, 32, 32, 1024, 256, 1, 1, 0),      # layer3 conv1: M = 12 288, 16 K-stages
    (12, 32, 32, 256, 256, 3, 1, 1),       # layer3 conv2: 36 K-stages
    (12, 32, 32, 256, 1024, 1, 1, 0),      # layer3 conv3 (and, transposed, the hand-over data gradient)
    (1, 1, 750000, 192, 64, 1, 1, 0),      # the stem GEMM over the im2col matrix: M = 750 000
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", REAL_CASES)
def test_conv_real_layer_shapes(dtype, case):
    """Forward, data gradient and weight gradient at the shapes the benchmark runs (auto tile / split-K), vs torch-CPU fp32 fed
    the same rounded operands.  Covers what the small cases cannot: multi-thousand-block grids, the XCD remap at grid sizes that
    are not multiples of 8, split-K chunking and the folded statistic rows at M = 187 500."""
    from tinyfaces import _hip, ops
    N, H, W, Cin, Cout, K, s, p = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    xr, wr = q(x, dtype).requires_grad_(True), q(w, dtype).requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=s, padding=p)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(q(gy, dtype))
    xd, gyd = to_nhwc(x, dtype), to_nhwc(gy, dtype)
    y, st = ops.conv2d_nhwc(xd, ops.pack_weight(w.cuda(), dtype), Cout, K, K, s, p, epi=_hip.EPI_STATS, want_stats=True)
    d = err(from_nhwc(y), ref.detach())
    n = ref.numel() / Cout
    d1 = err(st.sum(0).cpu()[0] / n, ref.detach().mean(dim=(0, 2, 3)))
    gx = ops.conv2d_nhwc(gyd, ops.pack_weight(w.cuda(), dtype, transpose=True), Cin, K, K, s, p, mode=1, out_hw=(H, W))
    dg = err(from_nhwc(gx), xr.grad)
    dw = ops.conv2d_wgrad(xd, gyd, Cin, Cout, K, K, s, p)
    dwe = err(dw.cpu(), wr.grad)
    report(f"conv_real[{dtype},{case}]", fwd_rel=d[2], mean_abs=d1[0], dgrad_rel=dg[2], wgrad_rel=dwe[2])
    assert d[2] < TOL[dtype] and d1[0] < 2e-3
    assert dg[2] < TOL[dtype]
    assert dwe[2] < (1e-4 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(2, 16, 16, 128, 64), (3, 32, 32, 1024, 512), (1, 63, 63, 128, 64)])
def test_scattered_downsample_gradient_accumulates_in_place(dtype, case):
    """r6: the data gradient of a 1x1 / stride-2 conv (downsample branch of layer2.0 / layer3.0) with aux == y: its rows are ADDED to the even-even
    pixels of a raster another launch already wrote -- no zero fill, no residual copy of the 2x larger raster --, with the hand-over epilogue
    riding along: ReLU mask of the previous block's output (MASK2) and the BN-backward sums (STATS3) taking only what THIS launch adds.
    Reference: torch conv_transpose2d on the same rounded operands; the raster holds an already masked gradient g1 and the sum rows hold g1's sums."""
    from tinyfaces import _hip, ops
    N, h, w_, Cgrad, Cin = case                       # gradient raster (N, h, w_, Cgrad) -> input raster (N, 2h-1, 2w_-1, Cin)
    H, W = 2 * h - 1, 2 * w_ - 1
    g = _g(hash(case) % 1000)
    wt = torch.randn(Cgrad, Cin, 1, 1, generator=g) / Cgrad ** 0.5          # the forward conv: Cin -> Cgrad, stride 2
    gy = torch.randn(N, Cgrad, h, w_, generator=g)
    yprev = torch.randn(N, Cin, H, W, generator=g)                            # the previous block's output: its sign is the ReLU mask
    x3 = torch.randn(N, Cin, H, W, generator=g)                               # its conv3 output (the BN-backward sum operand)
    mask = (q(yprev, dtype) > 0).float()
    g1 = q(torch.randn(N, Cin, H, W, generator=g), dtype) * mask              # what the hand-over conv left in the raster (masked)
    gds = F.conv_transpose2d(q(gy, dtype), q(wt, dtype), stride=2)            # (N, Cin, 2h-1, 2w_-1): nonzero on even-even pixels only
    ref = (g1 + gds) * mask
    raster = to_nhwc(g1, dtype)
    rows = _hip.lib().tf_get_stat_rows()
    sums0 = torch.stack([(g1 * 1.0).sum(dim=(0, 2, 3)), (g1 * q(x3, dtype)).sum(dim=(0, 2, 3))])          # (2, Cin): already counted by the hand-over
    stats = torch.zeros(rows, 2, Cin, device="cuda")
    stats[0] = sums0.cuda()
    wp = ops.pack_weight(wt.cuda(), dtype, transpose=True)
    out, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), wp, Cin, 1, 1, 2, 0, mode=1, out_hw=(H, W), epi=_hip.EPI_RES | _hip.EPI_MASK2 | _hip.EPI_STATS3,
                              aux=raster, aux2=to_nhwc(yprev, dtype), aux3=to_nhwc(x3, dtype), want_stats=True, out=raster, stats_into=stats)
    assert out.data_ptr() == raster.data_ptr()
    d = err(from_nhwc(out), ref)
    tot = st.sum(0).cpu()
    want = torch.stack([ref.sum(dim=(0, 2, 3)), (ref * q(x3, dtype)).sum(dim=(0, 2, 3))])
    # odd pixels must be untouched bit for bit
    odd = from_nhwc(out)[:, :, 1::2, :]
    report(f"scatter_inplace[{dtype},{case}]", rel=d[2], sum_rel=err(tot[0], want[0])[2], sumx_rel=err(tot[1], want[1])[2])
    assert torch.equal(odd, g1[:, :, 1::2, :].to(odd.dtype))
    assert d[2] < TOL[dtype]
    assert err(tot[0], want[0])[2] < 5e-3 and err(tot[1], want[1])[2] < 5e-3


WG_CASES = [
    # N, H, W, Cin, Cout, K, stride
    (1, 9, 7, 64, 64, 1, 1),
    (2, 33, 35, 128, 64, 3, 1),
    (2, 20, 22, 64, 64, 3, 1),
    (2, 21, 19, 128, 128, 3, 2),
    (3, 30, 30, 256, 1024, 1, 1),
    (2, 31, 29, 512, 1024, 1, 2),
    (2, 16, 16, 1024, 256, 1, 1),
    (4, 40, 40, 64, 256, 1, 1),
]


@pytest.mark.parametrize("tile", [0, 64, 128])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad(dtype, case, tile):
    from tinyfaces import ops
    N, H, W, Cin, Cout, K, s = case
    p = K // 2
    g = _g(hash(case) % 997)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, K, K, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(q(x, dtype), w, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(q(gy, dtype))
    dw = ops.conv2d_wgrad(to_nhwc(x, dtype), to_nhwc(gy, dtype), Cin, Cout, K, K, s, p, tile=tile)
    d = err(dw.cpu(), w.grad)
    report(f"wgrad[{dtype},{case},t{tile}]", maxabs=d[0], rel=d[2])
    assert d[2] < (5e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", DT)
def test_wgrad_with_prologue_and_padded_head(dtype):
    from tinyfaces import ops
    g = _g(77)
    N, H, W, Cin, Cout = 2, 12, 16, 128, 256
    raw = torch.randn(N, Cin, H, W, generator=g)
    ps, ph = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    act = q(torch.relu(q(raw, dtype) * ps.view(1, -1, 1, 1) + ph.view(1, -1, 1, 1)), dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(act, w, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(q(gy, dtype))
    dw = ops.conv2d_wgrad(to_nhwc(raw, dtype), to_nhwc(gy, dtype), Cin, Cout, 3, 3, 1, 1, pro=(ps.cuda(), ph.cuda(), True))
    d = err(dw.cpu(), w.grad)
    # head: Cout = 125 stored with lddy = 128
    x2 = torch.randn(N, 512, H, W, generator=g)
    w2 = (torch.randn(125, 512, 1, 1, generator=g) * 0.05).requires_grad_(True)
    y2 = F.conv2d(q(x2, dtype), w2)
    gy2 = torch.randn(y2.shape, generator=g)
    y2.backward(q(gy2, dtype))
    gy2p = torch.cat([gy2, torch.zeros(N, 3, H, W)], 1)
    dw2 = ops.conv2d_wgrad(to_nhwc(x2, dtype), to_nhwc(gy2p, dtype), 512, 125, 1, 1, 1, 0)
    x3 = torch.randn(N, 192, H, W, generator=g); x3[:, 147:] = 0          # stem im2col shape: Cin 147 stored with ldx 192
    w3 = (torch.randn(64, 147, 1, 1, generator=g) * 0.05).requires_grad_(True)
    y3 = F.conv2d(q(x3[:, :147], dtype), w3)
    gy3 = torch.randn(y3.shape, generator=g)
    y3.backward(q(gy3, dtype))
    dw3 = ops.conv2d_wgrad(to_nhwc(x3, dtype), to_nhwc(gy3, dtype), 147, 64, 1, 1, 1, 0)
    d3 = err(dw3.cpu(), w3.grad)
    assert d3[2] < (5e-5 if dtype == torch.float32 else 3e-3)
    d2 = err(dw2.cpu(), w2.grad)
    report(f"wgrad_pro_head[{dtype}]", rel=d[2], head_rel=d2[2])
    tol = 5e-5 if dtype == torch.float32 else 3e-3
    assert d[2] < tol and d2[2] < tol


# ------------------------------------------------------------------ HBM-bound companions
@pytest.mark.parametrize("dtype", DT)
def test_stem_im2col_and_maxpool(dtype, hip):
    from tinyfaces import ops
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    g = _g(5)
    N, H, W = 2, 37, 45
    x = torch.randn(N, 3, H, W, generator=g)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    col = torch.empty(N * OH * OW, 192, dtype=dtype, device="cuda")
    xd = x.cuda()
    assert lib().tf_stem_im2col(ptr(xd), N, H, W, tf_dtype(dtype), ptr(col), 192, stream()) == 0
    ref = F.unfold(q(x, dtype), 7, padding=3, stride=2).transpose(1, 2).reshape(-1, 147)      # k = c*49 + kh*7 + kw
    d = err(col.float().cpu()[:, :147], ref)
    assert d[0] == 0 and float(col.float().abs()[:, 147:].max()) == 0
    # maxpool (+ fused BN/ReLU prologue, arg-max) and its backward
    C = 64
    a = torch.randn(N, C, OH, OW, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    act = torch.relu(q(a, dtype) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    pref = F.max_pool2d(act, 3, 2, 1)
    PH, PW = pref.shape[2:]
    xin = to_nhwc(a, dtype)
    y = torch.empty(N, PH, PW, C, dtype=dtype, device="cuda")
    idx = torch.empty(N * PH * PW * C, dtype=torch.uint8, device="cuda")
    scd, shd = sc.cuda(), sh.cuda()
    assert lib().tf_maxpool_fwd(tf_dtype(dtype), ptr(xin), N, OH, OW, C, ptr(scd), ptr(shd), ptr(y), ptr(idx), stream()) == 0
    d = err(from_nhwc(y), q(pref.detach(), dtype))
    gp = torch.randn(pref.shape, generator=g)
    pref.backward(q(gp, dtype))
    gz = torch.empty_like(xin)
    gpd = to_nhwc(gp, dtype)
    assert lib().tf_maxpool_bwd(tf_dtype(dtype), ptr(gpd), ptr(idx), ptr(xin), ptr(scd), ptr(shd), N, OH, OW, C, ptr(gz), stream()) == 0
    # reference: gradient w.r.t. the activation, then through the ReLU mask
    gref = act.grad * (act.detach() > 0)
    d2 = err(from_nhwc(gz), gref)
    report(f"im2col_maxpool[{dtype}]", pool_maxabs=d[0], pool_bwd_rel=d2[2])
    assert d[0] < (1e-6 if dtype == torch.float32 else 1e-2) and d2[2] < (1e-6 if dtype == torch.float32 else 8e-3)
    # r4: the same backward with the column sums of the stem's BN backward taken in the same pass: gz bit-equal, sums = those of the stored gz
    import ctypes as C_
    gz2 = torch.empty_like(xin)
    rows = C_.c_int(0)
    st = torch.zeros(hip.lib().tf_get_stat_rows(), 2, C, device="cuda")
    assert lib().tf_maxpool_bwd_stats(tf_dtype(dtype), ptr(gpd), ptr(idx), ptr(xin), ptr(scd), ptr(shd), N, OH, OW, C, ptr(gz2), ptr(st), C_.byref(rows), stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(gz2, gz) and 1 <= rows.value <= st.shape[0]
    s = st[:rows.value].sum(0).cpu().double()
    gzf, xf = gz.double().cpu().reshape(-1, C), xin.double().cpu().reshape(-1, C)
    e1, e2 = err(s[0], gzf.sum(0)), err(s[1], (gzf * xf).sum(0))
    report(f"maxpool_bwd_stats[{dtype}]", rows=rows.value, sum_rel=e1[2], sumx_rel=e2[2])
    assert e1[2] < 1e-5 and e2[2] < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 37, 45), (1, 130, 71), (3, 64, 64)])
def test_stem_conv_direct_equals_im2col_gemm_and_torch(dtype, shape):
    """tf_stem_conv (r4, csrc/stem_conv.hip): conv1 straight from the NCHW fp32 image -- against torch's conv on the rounded operands, against
    the im2col + GEMM path it replaces (same products, another summation order), its batch statistics against the sums of the fp32 result,
    and the folded-BN + ReLU epilogue of the evaluation graph; image sizes with partial tiles on every border, several tiles per block."""
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    g = _g(31)
    N, H, W = shape
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    ref = F.conv2d(q(x, dtype), q(w, dtype), stride=2, padding=3)                      # fp32 accumulation of the rounded operands
    y, st = ops.stem_conv(x.cuda(), w.cuda(), dtype, epi=_hip.EPI_STATS)
    d = err(from_nhwc(y), q(ref, dtype))
    s = st.sum(0).cpu().double()
    d1, d2 = err(s[0], ref.double().sum(dim=(0, 2, 3))), err(s[1], (ref.double() ** 2).sum(dim=(0, 2, 3)))
    # the path of rounds 1-3
    OH, OW = ref.shape[2:]
    col = torch.empty(N * OH * OW, 192, dtype=dtype, device="cuda")
    xd = x.cuda()
    assert lib().tf_stem_im2col(ptr(xd), N, H, W, tf_dtype(dtype), ptr(col), 192, stream()) == 0
    wp = ops.pack_weight(w.cuda().reshape(64, 147, 1, 1), dtype, cols_pad=192)
    y_old = ops.conv2d_nhwc(col.view(1, 1, N * OH * OW, 192), wp, 64, 1, 1, 1, 0)
    d_old = err(y.float().cpu().reshape(-1, 64), y_old.float().cpu().reshape(-1, 64))
    # evaluation epilogue
    sc, sh = (torch.rand(64, generator=g) + 0.5), torch.randn(64, generator=g) * 0.2
    y2 = ops.stem_conv(x.cuda(), w.cuda(), dtype, epi=_hip.EPI_AFFINE | _hip.EPI_RELU, scale=sc.cuda(), shift=sh.cuda())
    d3 = err(from_nhwc(y2), q(torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), dtype))
    y0 = ops.stem_conv(x.cuda(), w.cuda(), dtype)
    report(f"stem_conv[{dtype},{shape}]", rel=d[2], vs_im2col=d_old[2], sum_rel=d1[2], sumsq_rel=d2[2], eval_rel=d3[2])
    assert torch.equal(y0, y)
    tol = {torch.bfloat16: 6e-3, torch.float16: 1e-3}[dtype]
    assert d[2] < tol and d_old[2] < tol and d3[2] < tol and d1[2] < 1e-4 and d2[2] < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 37, 45), (1, 130, 71), (3, 64, 64), (2, 250, 250)])
def test_stem_wgrad_direct_equals_autograd(dtype, shape):
    """tf_stem_wgrad (r4): the weight gradient of conv1 straight from the image against torch's autograd on the rounded operands (fp32
    accumulation), partial tiles on every border, blocks that walk several tiles, more tiles than blocks (250x250)."""
    from tinyfaces import ops
    g = _g(37)
    N, H, W = shape
    x = torch.randn(N, 3, H, W, generator=g)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    gy = torch.randn(N, 64, OH, OW, generator=g)
    w = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(q(x, dtype), w, stride=2, padding=3).backward(q(gy, dtype))
    dw = ops.stem_wgrad(x.cuda(), to_nhwc(gy, dtype))
    d = err(dw.cpu(), w.grad)
    # ... and with the stem's BN-backward apply folded into the gradient operand: cA * g + cB * x_conv + cD, rounded to the operand type
    xc = torch.randn(N, 64, OH, OW, generator=g)
    cA, cB, cD = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.05
    ga = q(cA.view(1, -1, 1, 1) * q(gy, dtype) + cB.view(1, -1, 1, 1) * q(xc, dtype) + cD.view(1, -1, 1, 1), dtype)
    w2 = torch.zeros(64, 3, 7, 7, requires_grad=True)
    F.conv2d(q(x, dtype), w2, stride=2, padding=3).backward(ga)
    dw2 = ops.stem_wgrad(x.cuda(), to_nhwc(gy, dtype), to_nhwc(xc, dtype), cA.cuda(), cB.cuda(), cD.cuda())
    d2 = err(dw2.cpu(), w2.grad)
    report(f"stem_wgrad[{dtype},{shape}]", rel=d[2], max_ref=d[1], applied_rel=d2[2])
    assert d[2] < 2e-4 and d2[2] < 2e-3          # (the applied operand is re-rounded: one ulp of the operand type where the fp32 orders differ)


@pytest.mark.parametrize("dtype", DT)
def test_bn_train_forward_backward_chain(dtype):
    """colstats -> bn_finalize -> bn_add_relu forward; colstats(masked) -> bn_bwd_finalize -> bn_bwd_apply backward,
    vs torch BatchNorm2d(train) + add + relu autograd."""
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    g = _g(9)
    N, H, W, C = 3, 11, 13, 256
    M = N * H * W
    tfd = tf_dtype(dtype)
    xr = torch.randn(N, C, H, W, generator=g) * 1.3 + 0.2
    idn = torch.randn(N, C, H, W, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    xq = q(xr, dtype).requires_grad_(True)
    gm = gamma.clone().requires_grad_(True); bt = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    rmd, rvd = rm.clone().cuda(), rv.clone().cuda()          # before torch updates rm / rv in place
    yref = torch.relu(F.batch_norm(xq, rm, rv, gm, bt, True, 0.1, 1e-5) + q(idn, dtype))
    gy = torch.randn(yref.shape, generator=g)
    yref.backward(q(gy, dtype))
    x_d, id_d = to_nhwc(xr, dtype), to_nhwc(idn, dtype)
    nb = lib().tf_colstats_blocks(M, C, tfd)
    part = torch.zeros(nb, 2, C, device="cuda")
    assert lib().tf_colstats(tfd, ptr(x_d), None, ptr(x_d), None, M, C, C, ptr(part), stream()) == 0
    bufs = [torch.zeros(C, device="cuda") for _ in range(9)]
    scale, shift, mean, invstd, dga, dbe, cA, cB, cD = bufs
    gd, bd = gamma.cuda(), beta.cuda()
    assert lib().tf_bn_finalize(ptr(part), nb, C, C, float(M), ptr(gd), ptr(bd), 1e-5, 0.1, ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                ptr(rmd), ptr(rvd), 0, stream()) == 0
    y_d = torch.empty_like(x_d)
    assert lib().tf_bn_add_relu(tfd, ptr(x_d), ptr(scale), ptr(shift), ptr(id_d), None, None, M, C, ptr(y_d), stream()) == 0
    dy = err(from_nhwc(y_d), yref.detach())
    drm, drv = err(rmd.cpu(), rm), err(rvd.cpu(), rv)
    gy_d = to_nhwc(gy, dtype)
    part3 = torch.zeros(nb, 2, C, device="cuda")
    assert lib().tf_colstats(tfd, ptr(gy_d), ptr(y_d), ptr(x_d), None, M, C, C, ptr(part3), stream()) == 0
    assert lib().tf_bn_bwd_finalize(ptr(part3), nb, 2, 1, C, C, float(M), ptr(gd), ptr(mean), ptr(invstd), ptr(dga), ptr(dbe), ptr(cA),
                                    ptr(cB), ptr(cD), 0, stream()) == 0
    gx_d = torch.empty_like(x_d)
    assert lib().tf_bn_bwd_apply(tfd, ptr(gy_d), ptr(y_d), ptr(x_d), ptr(cA), ptr(cB), ptr(cD), M, C, ptr(gx_d), stream()) == 0
    dgx = err(from_nhwc(gx_d), xq.grad)
    dgg, dgb = err(dga.cpu(), gm.grad), err(dbe.cpu(), bt.grad)
    report(f"bn_chain[{dtype}]", y_rel=dy[2], rm=drm[0], rv=drv[0], gx_rel=dgx[2], dgamma_rel=dgg[2], dbeta_rel=dgb[2])
    t = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert dy[2] < t and dgx[2] < (1e-4 if dtype == torch.float32 else 2e-2)
    assert drm[0] < 1e-4 and drv[0] < 1e-3 and dgg[2] < (1e-4 if dtype == torch.float32 else 2e-2) and dgb[2] < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,ds", [(64, False), (256, True), (1024, False), (512, True)])
def test_bn_fused_consumers_chain(dtype, C, ds):
    """The in-kernel-finalize consumers (bn_fused.hip): colstats rows -> tf_bn_relu_fused / tf_bn_add_relu_fused forward,
    colstats(masked) rows -> tf_bn_bwd_apply_fused backward, vs torch BatchNorm2d(train) (+ second BN on the residual
    branch when ds) + add + relu autograd.  Rows come from a zeroed region, as in the executor."""
    from tinyfaces import _hip
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    g = _g(100 + C)
    N, H, W = 3, 11, 13
    M = N * H * W
    tfd = tf_dtype(dtype)
    R = lib().tf_get_stat_rows()
    assert 1 <= R <= 16
    xr = torch.randn(N, C, H, W, generator=g) * 1.3 + 0.2
    idn = torch.randn(N, C, H, W, generator=g) * 0.7 - 0.1
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    gamma2, beta2 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    xq = q(xr, dtype).requires_grad_(True)
    iq = q(idn, dtype).requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    gm2, bt2 = gamma2.clone().requires_grad_(True), beta2.clone().requires_grad_(True)
    rm, rv, rm2, rv2 = torch.zeros(C), torch.ones(C), torch.zeros(C), torch.ones(C)
    dev = lambda t: t.clone().cuda()
    rmd, rvd, rm2d, rv2d = dev(rm), dev(rv), dev(rm2), dev(rv2)
    branch = F.batch_norm(iq, rm2, rv2, gm2, bt2, True, 0.1, 1e-5) if ds else iq
    yref = torch.relu(F.batch_norm(xq, rm, rv, gm, bt, True, 0.1, 1e-5) + branch)
    r1ref = torch.relu(F.batch_norm(xq.detach(), torch.zeros(C), torch.ones(C), gamma, beta, True, 0.1, 1e-5))
    gy = torch.randn(yref.shape, generator=g)
    yref.backward(q(gy, dtype))
    x_d, id_d = to_nhwc(xr, dtype), to_nhwc(idn, dtype)
    z = lambda: torch.zeros(C, device="cuda")

    def fwd_rows(t):
        rows = torch.zeros(16, 2, C, device="cuda")
        assert lib().tf_colstats(tfd, ptr(t), None, ptr(t), None, M, C, C, ptr(rows), stream()) == 0
        return rows
    rows1, rows2 = fwd_rows(x_d), fwd_rows(id_d)
    gd, bd, g2d, b2d = gamma.cuda(), beta.cuda(), gamma2.cuda(), beta2.cuda()
    sc, sh, mean, invstd = z(), z(), z(), z()
    sc2, sh2, mean2, invstd2 = z(), z(), z(), z()
    d1 = _hip.BnFwdDesc(ptr(rows1), ptr(gd), ptr(bd), ptr(sc), ptr(sh), ptr(mean), ptr(invstd), ptr(rmd), ptr(rvd))
    d2 = _hip.BnFwdDesc(ptr(rows2), ptr(g2d), ptr(b2d), ptr(sc2), ptr(sh2), ptr(mean2), ptr(invstd2), ptr(rm2d), ptr(rv2d))
    # bn + relu (no running-stat update in this call: the descriptor without running pointers)
    keep = [z(), z(), z(), z()]                                   # published vectors of the throw-away call (must stay alive)
    d1n = _hip.BnFwdDesc(ptr(rows1), ptr(gd), ptr(bd), ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), None, None)
    r1_d = torch.empty_like(x_d)
    assert lib().tf_bn_relu_fused(tfd, ptr(x_d), d1n, R, M, C, float(M), 1e-5, 0.1, ptr(r1_d), stream()) == 0
    dr1 = err(from_nhwc(r1_d), r1ref)
    y_d = torch.empty_like(x_d)
    assert lib().tf_bn_add_relu_fused(tfd, ptr(x_d), d1, ptr(id_d), d2 if ds else None, R, M, C, float(M), 1e-5, 0.1, ptr(y_d), stream()) == 0
    dy = err(from_nhwc(y_d), yref.detach())
    drm, drv = err(rmd.cpu(), rm), err(rvd.cpu(), rv)
    if ds:
        drm2, drv2 = err(rm2d.cpu(), rm2), err(rv2d.cpu(), rv2)
        assert drm2[0] < 1e-4 and drv2[0] < 1e-3
    # backward: sums of gz = gy*(y>0): k1 with x, k2 with the residual-branch input (ds)
    gy_d = to_nhwc(gy, dtype)
    nk = 3 if ds else 2
    brow = torch.zeros(16, nk, C, device="cuda")
    assert lib().tf_colstats(tfd, ptr(gy_d), ptr(y_d), ptr(x_d), ptr(id_d) if ds else None, M, C, C, ptr(brow), stream()) == 0
    dga, dbe, dga2, dbe2 = z(), z(), z(), z()
    b1 = _hip.BnBwdDesc(ptr(brow), ptr(gd), ptr(mean), ptr(invstd), ptr(dga), ptr(dbe), nk, 1)
    gx_d = torch.empty_like(x_d)
    assert lib().tf_bn_bwd_apply_fused(tfd, ptr(gy_d), ptr(y_d), ptr(x_d), b1, R, M, C, float(M), ptr(gx_d), stream()) == 0
    dgx = err(from_nhwc(gx_d), xq.grad)
    dgg, dgb = err(dga.cpu(), gm.grad), err(dbe.cpu(), bt.grad)
    t = 1e-5 if dtype == torch.float32 else 1.5e-2
    tg = 1e-4 if dtype == torch.float32 else 2e-2
    if ds:
        b2 = _hip.BnBwdDesc(ptr(brow), ptr(g2d), ptr(mean2), ptr(invstd2), ptr(dga2), ptr(dbe2), nk, 2)
        gi_d = torch.empty_like(x_d)
        assert lib().tf_bn_bwd_apply_fused(tfd, ptr(gy_d), ptr(y_d), ptr(id_d), b2, R, M, C, float(M), ptr(gi_d), stream()) == 0
        dgi = err(from_nhwc(gi_d), iq.grad)
        assert dgi[2] < tg and err(dga2.cpu(), gm2.grad)[2] < tg and err(dbe2.cpu(), bt2.grad)[2] < tg
    report(f"bn_fused[{dtype},C={C},ds={ds}]", r1_rel=dr1[2], y_rel=dy[2], rm=drm[0], rv=drv[0], gx_rel=dgx[2], dgamma_rel=dgg[2], dbeta_rel=dgb[2])
    assert dr1[2] < t and dy[2] < t and dgx[2] < tg
    assert drm[0] < 1e-4 and drv[0] < 1e-3 and dgg[2] < tg and dgb[2] < tg


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("hw", [(8, 8), (13, 17), (63, 63), (30, 40), (6, 10)])
def test_upsample_add_crop_fwd_bwd(dtype, hw):
    """score4_upsample (ConvTranspose2d k4 s2 p1, bilinear diagonal) + crop + add, vs torch (model.py:104-126)."""
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    from oracle.model import bilinear_kernel
    g = _g(hw[0])
    B, C = 2, 125
    H3, W3 = hw
    H4, W4 = (H3 + 1) // 2, (W3 + 1) // 2
    s3 = torch.randn(B, C, H3, W3, generator=g)
    s4 = torch.randn(B, C, H4, W4, generator=g, requires_grad=False)
    wfull = torch.zeros(C, C, 4, 4)
    wfull[torch.arange(C), torch.arange(C)] = torch.from_numpy(bilinear_kernel(4)).float()
    s4q = q(s4, dtype).requires_grad_(True)
    s3q = q(s3, dtype).requires_grad_(True)
    up = F.conv_transpose2d(s4q, wfull, stride=2, padding=1)[:, :, :H3, :W3]
    ref = s3q + up
    pad = lambda t: torch.cat([t, torch.zeros(B, 3, *t.shape[2:])], 1)
    s3d, s4d = to_nhwc(pad(s3), dtype), to_nhwc(pad(s4), dtype)
    diag = wfull[torch.arange(C), torch.arange(C)].reshape(C, 16).contiguous().cuda()
    out = torch.empty(B, C, H3, W3, device="cuda")
    tfd = tf_dtype(dtype)
    assert lib().tf_upsample_add_crop(tfd, ptr(s3d), ptr(s4d), ptr(diag), B, C, 128, H3, W3, H4, W4, ptr(out), stream()) == 0
    d = err(out.cpu(), ref.detach())
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    g3 = torch.empty(B, H3, W3, 128, dtype=dtype, device="cuda")
    g4 = torch.empty(B, H4, W4, 128, dtype=dtype, device="cuda")
    god = go.cuda()
    assert lib().tf_upsample_add_crop_bwd(tfd, ptr(god), ptr(diag), B, C, 128, H3, W3, H4, W4, ptr(g3), ptr(g4), stream()) == 0
    d3 = err(from_nhwc(g3)[:, :C], s3q.grad)
    d4 = err(from_nhwc(g4)[:, :C], s4q.grad)
    report(f"upsample[{dtype},{hw}]", fwd_maxabs=d[0], g3=d3[2], g4=d4[2])
    assert d[0] < 1e-5 and d3[2] < TOL[dtype] and d4[2] < TOL[dtype]
    assert float(from_nhwc(g3)[:, C:].abs().max()) == 0


# ------------------------------------------------------------------ 32x32x16-fragment tiles of the LDS-DMA kernel (round 2)
MMA32_TILES = [14, 15, 16, 34, 35, 36, 44, 45, 46]       # 128x128 / 128x64 / 64x128, ring depth 3 / none / 2
HALF = [torch.bfloat16, torch.float16]
TOL_H = {torch.bfloat16: 6e-3, torch.float16: 1e-3}


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", CONV_CASES + [(2, 40, 44, 256, 256, 3, 1, 1), (1, 50, 50, 1024, 256, 1, 1, 0)])
@pytest.mark.parametrize("tile", MMA32_TILES)
def test_conv_mma32_forward_and_dgrad(dtype, case, tile):
    """Forward and data gradient on 32x32x16 fragments (64x64 / 64x32 / 32x64 wave tiles): every gather kind, edge tiles in M
    and N, ring depths 1-3, against torch-CPU fp32 fed the same rounded operands."""
    from tinyfaces import ops
    N, H, W, Cin, Cout, K, s, p = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    xr = q(x, dtype).requires_grad_(True)
    ref = F.conv2d(xr, q(w, dtype), stride=s, padding=p)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(q(gy, dtype))
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, K, K, s, p, tile=tile)
    d = err(from_nhwc(y)[:, :Cout], ref.detach())
    gx = ops.conv2d_nhwc(to_nhwc(gy, dtype), ops.pack_weight(w.cuda(), dtype, transpose=True), Cin, K, K, s, p, mode=1, out_hw=(H, W), tile=tile)
    dg = err(from_nhwc(gx), xr.grad)
    report(f"conv_mma32[{dtype},{case},t{tile}]", fwd_rel=d[2], dgrad_rel=dg[2])
    assert d[2] < TOL_H[dtype] and dg[2] < TOL_H[dtype]


@pytest.mark.parametrize("tile", MMA32_TILES)
def test_conv_mma32_epilogues(tile):
    """Every epilogue of the executor on the new tiles: STATS (training forward), AFFINE+RES+RELU with a padded head (eval),
    MASK+STATS2 and JOIN (data gradients), RES+MASK2+STATS3 (the hand-over)."""
    from tinyfaces import _hip, ops
    dtype = torch.bfloat16
    g = _g(41 + tile)
    # STATS
    N, H, W, Cin, Cout = 2, 19, 23, 256, 128
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    ref = F.conv2d(q(x, dtype), q(w, dtype), padding=1)
    y, st = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, 3, 3, 1, 1, epi=_hip.EPI_STATS, want_stats=True, tile=tile)
    n = ref.numel() / Cout
    ssum = st.sum(0).cpu()
    assert err(from_nhwc(y), ref)[2] < TOL[dtype]
    assert err(ssum[0] / n, ref.mean(dim=(0, 2, 3)))[0] < 2e-3 and err(ssum[1] / n, (ref ** 2).mean(dim=(0, 2, 3)))[2] < 2e-3
    # eval epilogue, Cout = 125 in 128 columns
    N, H, W, Cin, Cout = 2, 13, 17, 512, 125
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(N, 128, H, W, generator=g)
    ref = torch.relu(F.conv2d(q(x, dtype), q(w, dtype)) * sc[:Cout].view(1, -1, 1, 1) + sh[:Cout].view(1, -1, 1, 1) + q(res, dtype)[:, :Cout])
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), ops.pack_weight(w.cuda(), dtype), Cout, 1, 1, 1, 0, ldy=128,
                        epi=_hip.EPI_AFFINE | _hip.EPI_RES | _hip.EPI_RELU, epi_scale=sc.cuda(), epi_shift=sh.cuda(), aux=to_nhwc(res, dtype), tile=tile)
    assert err(from_nhwc(y)[:, :Cout], ref)[2] < TOL[dtype]
    # MASK + STATS2, JOIN
    N, H, W, C1, C2 = 2, 16, 18, 256, 64
    gy = torch.randn(N, C1, H, W, generator=g)
    w = torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5
    craw = torch.randn(N, C2, H, W, generator=g)
    ms, mh = torch.rand(C2, generator=g) + 0.5, torch.randn(C2, generator=g) * 0.5
    base = F.conv_transpose2d(q(gy, dtype), q(w, dtype))
    ref = base * ((q(craw, dtype) * ms.view(1, -1, 1, 1) + mh.view(1, -1, 1, 1)) > 0)
    wt = ops.pack_weight(w.cuda(), dtype, transpose=True)
    y, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), wt, C2, 1, 1, 1, 0, mode=1, out_hw=(H, W), epi=_hip.EPI_MASK | _hip.EPI_STATS2,
                            aux=to_nhwc(craw, dtype), mask=(ms.cuda(), mh.cuda()), want_stats=True, tile=tile)
    s = st.sum(0).cpu()
    assert err(from_nhwc(y), ref)[2] < TOL[dtype]
    assert err(s[0], ref.sum(dim=(0, 2, 3)))[2] < 5e-3 and err(s[1], (ref * q(craw, dtype)).sum(dim=(0, 2, 3)))[2] < 5e-3
    y2, g3 = torch.randn(N, C2, H, W, generator=g), torch.randn(N, C2, H, W, generator=g)
    yj = ops.conv2d_nhwc(to_nhwc(gy, dtype), wt, C2, 1, 1, 1, 0, mode=1, out_hw=(H, W), epi=_hip.EPI_JOIN, aux2=to_nhwc(y2, dtype),
                         aux3=to_nhwc(g3, dtype), tile=tile)
    assert err(from_nhwc(yj), base + q(g3, dtype) * (q(y2, dtype) > 0))[2] < TOL[dtype]
    # hand-over: RES + MASK2 + STATS3
    N, H, W, C1, C2 = 2, 17, 15, 64, 256
    gy = torch.randn(N, C1, H, W, generator=g)
    w = torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5
    res, yprev = torch.randn(N, C2, H, W, generator=g), torch.randn(N, C2, H, W, generator=g)
    c3 = torch.randn(N, C2, H, W, generator=g) * 1.5 + 0.3
    ref = (F.conv_transpose2d(q(gy, dtype), q(w, dtype)) + q(res, dtype)) * (q(yprev, dtype) > 0)
    y, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), ops.pack_weight(w.cuda(), dtype, transpose=True), C2, 1, 1, 1, 0, mode=1, out_hw=(H, W),
                            epi=_hip.EPI_RES | _hip.EPI_MASK2 | _hip.EPI_STATS3, aux=to_nhwc(res, dtype), aux2=to_nhwc(yprev, dtype),
                            aux3=to_nhwc(c3, dtype), want_stats=True, tile=tile)
    s = st.sum(0).cpu()
    assert err(from_nhwc(y), ref)[2] < TOL[dtype]
    assert err(s[0], ref.sum(dim=(0, 2, 3)))[2] < 5e-3 and err(s[1], (ref * q(c3, dtype)).sum(dim=(0, 2, 3)))[2] < 5e-3


def test_conv_mma32_refuses_fp32():
    from tinyfaces import ops
    x = torch.randn(1, 8, 8, 64, device="cuda")
    w = ops.pack_weight(torch.randn(64, 64, 1, 1, device="cuda"), torch.float32)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, w, 64, 1, 1, 1, 0, tile=14)


# ------------------------------------------------------------------ halo-resident 3x3 / stride 1 kernel (round 2, csrc/conv3x3h.hip)
C3H_CASES = [
    # N, H, W, Cin, Cout
    (1, 1, 1, 64, 128),         # one pixel: 8 of the 9 taps read only the zero halo
    (2, 4, 32, 64, 128),        # exactly one tile per image
    (2, 5, 33, 128, 128),       # one row / one column past a tile: edge tiles in both directions, 2 channel chunks
    (1, 21, 70, 64, 256),       # 3 column tiles (the last 6 wide), 2 channel tiles
    (3, 32, 32, 256, 256),      # layer-3 shape (smaller batch)
    (1, 63, 63, 128, 128),      # layer-2 shape
    (1, 9, 40, 320, 384),       # 5 chunks (odd: the one-tap stages), 3 channel tiles
    (1, 12, 40, 512, 128),      # 8 chunks = 4 two-chunk bodies of the paired-tap loop (r5), a ragged last column tile
]


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", C3H_CASES)
def test_conv3x3h_forward_and_dgrad(dtype, case):
    """Forward and data gradient of the halo-resident kernel (tile code 50): chunk-major K order, taps as row shifts of the
    6 x 34 frame, K halves added in the epilogue -- against torch-CPU fp32 fed the same rounded operands, and against the
    im2col kernel (tile 13) on the same operands (same products, different summation order)."""
    from tinyfaces import ops
    N, H, W, Cin, Cout = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xr = q(x, dtype).requires_grad_(True)
    ref = F.conv2d(xr, q(w, dtype), padding=1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(q(gy, dtype))
    wp, wt = ops.pack_weight(w.cuda(), dtype), ops.pack_weight(w.cuda(), dtype, transpose=True)
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, tile=50)
    y13 = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, tile=13)
    d = err(from_nhwc(y), ref.detach())
    gx = ops.conv2d_nhwc(to_nhwc(gy, dtype), wt, Cin, 3, 3, 1, 1, mode=1, out_hw=(H, W), tile=50) if Cin % 128 == 0 else None
    report(f"conv3x3h[{dtype},{case}]", fwd_rel=d[2], vs_im2col=err(from_nhwc(y), from_nhwc(y13).float().cpu())[2])
    assert d[2] < TOL_H[dtype]
    assert err(from_nhwc(y), from_nhwc(y13).float().cpu())[2] < TOL_H[dtype]
    if gx is not None:                      # the data gradient's output channels are the forward's input channels: 128-multiples only
        dg = err(from_nhwc(gx), xr.grad)
        assert dg[2] < TOL_H[dtype]


def test_conv3x3h_epilogues():
    """The epilogues the executor asks of a 3x3 conv: STATS (training forward; tile pixels outside the image must not count),
    AFFINE + RELU (evaluation), MASK + STATS2 (data gradient through ReLU(BN)), with more tiles than statistic rows."""
    from tinyfaces import _hip, ops
    dtype = torch.bfloat16
    g = _g(77)
    N, H, W, Cin, Cout = 3, 30, 45, 128, 256          # 3 * 8 * 2 = 48 tiles > TF_STAT_ROWS: atomics into mt % rows
    x = torch.randn(N, Cin, H, W, generator=g) + 0.25
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    ref = F.conv2d(q(x, dtype), q(w, dtype), padding=1)
    wp = ops.pack_weight(w.cuda(), dtype)
    y, st = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, epi=_hip.EPI_STATS, want_stats=True, tile=50)
    n = ref.numel() / Cout
    ssum = st.sum(0).cpu()
    assert st.shape[0] < 48          # folded into tf_get_stat_rows() rows
    assert err(from_nhwc(y), ref)[2] < TOL[dtype]
    assert err(ssum[0] / n, ref.mean(dim=(0, 2, 3)))[0] < 2e-3 and err(ssum[1] / n, (ref ** 2).mean(dim=(0, 2, 3)))[2] < 2e-3
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    y = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, epi=_hip.EPI_AFFINE | _hip.EPI_RELU, epi_scale=sc.cuda(), epi_shift=sh.cuda(), tile=50)
    assert err(from_nhwc(y), torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))[2] < TOL[dtype]
    # data gradient: dX = conv_transpose(dY) masked by the producer's ReLU(BN(c_raw)), with the dgamma / dbeta partial sums
    gy = torch.randn(N, Cout, H, W, generator=g)
    craw = torch.randn(N, Cin, H, W, generator=g)
    ms, mh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.5
    base = F.conv_transpose2d(q(gy, dtype), q(w, dtype), padding=1)
    refm = base * ((q(craw, dtype) * ms.view(1, -1, 1, 1) + mh.view(1, -1, 1, 1)) > 0)
    wt = ops.pack_weight(w.cuda(), dtype, transpose=True)
    y, st = ops.conv2d_nhwc(to_nhwc(gy, dtype), wt, Cin, 3, 3, 1, 1, mode=1, out_hw=(H, W), epi=_hip.EPI_MASK | _hip.EPI_STATS2,
                            aux=to_nhwc(craw, dtype), mask=(ms.cuda(), mh.cuda()), want_stats=True, tile=50)
    s = st.sum(0).cpu()
    assert err(from_nhwc(y), refm)[2] < TOL[dtype]
    assert err(s[0], refm.sum(dim=(0, 2, 3)))[2] < 5e-3 and err(s[1], (refm * q(craw, dtype)).sum(dim=(0, 2, 3)))[2] < 5e-3


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("case", [(1, 120, 160, 256, 256), (1, 118, 157, 128, 256), (2, 61, 150, 64, 256)])
def test_conv3x3h_six_row_tiles_of_the_big_pyramid_levels(dtype, case):
    """r6: an evaluation launch (folded BN + ReLU) whose 4-row tiles would need a second, nearly empty round of blocks takes 6-row tiles
    (csrc/conv3x3h.hip: Geo<6>; 1 x 120 x 160 = the 1920 x 2560 pyramid level: 300 blocks -> 200).  Same products in the same order per
    output pixel, so the result must equal the 4-row instantiation of the SAME kernel bit for bit -- taken here from the data-gradient-free
    STATS-less path that never uses tall tiles (no epilogue: plain store) after undoing the affine -- and torch-CPU within the usual bar.
    Ragged cases: a last row tile of 4 of 6 rows, a last column tile of 29 / 22 of 32 columns, two images."""
    from tinyfaces import _hip, ops
    N, H, W, Cin, Cout = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    ref = torch.relu(F.conv2d(q(x, dtype), q(w, dtype), padding=1))
    wp = ops.pack_weight(w.cuda(), dtype)
    one, zero = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    y6 = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, epi=_hip.EPI_AFFINE | _hip.EPI_RELU, epi_scale=one, epi_shift=zero, tile=50)
    assert _hip.lib().tf_debug_conv3x3h_tile_rows() == 6          # the tall instantiation really ran
    y4 = ops.conv2d_nhwc(to_nhwc(x, dtype), wp, Cout, 3, 3, 1, 1, tile=50)                      # no epilogue flags: 4-row tiles
    assert _hip.lib().tf_debug_conv3x3h_tile_rows() == 4
    d = err(from_nhwc(y6), ref)
    report(f"conv3x3h_tall[{dtype},{case}]", rel=d[2], bit_identical=int(torch.equal(y6, torch.relu(y4))))
    assert d[2] < TOL_H[dtype]
    assert torch.equal(y6, torch.relu(y4))                         # x * 1 + 0 and ReLU are exact: the two tile heights agree bit for bit
    # a launch that fills whole rounds either way keeps the 4-row tile (layer 3 of the training batch: 192 blocks)
    xs = torch.randn(3, 32, 32, Cin, device="cuda").to(dtype)
    ops.conv2d_nhwc(xs, wp, Cout, 3, 3, 1, 1, epi=_hip.EPI_AFFINE | _hip.EPI_RELU, epi_scale=one, epi_shift=zero, tile=50)
    assert _hip.lib().tf_debug_conv3x3h_tile_rows() == 4


def test_conv3x3h_refuses_what_it_cannot_do():
    from tinyfaces import ops
    x = torch.randn(1, 8, 8, 64, device="cuda", dtype=torch.bfloat16)
    w = ops.pack_weight(torch.randn(64, 64, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(RuntimeError):                                   # 64 output channels: not a multiple of the 128-wide tile
        ops.conv2d_nhwc(x, w, 64, 3, 3, 1, 1, tile=50)
    w = ops.pack_weight(torch.randn(128, 64, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(RuntimeError):                                   # stride 2
        ops.conv2d_nhwc(x, w, 128, 3, 3, 2, 1, tile=50)
    with pytest.raises(RuntimeError):                                   # fp32
        ops.conv2d_nhwc(x.float(), ops.pack_weight(torch.randn(128, 64, 3, 3, device="cuda"), torch.float32), 128, 3, 3, 1, 1, tile=50)


# ------------------------------------------------------------------ all-taps 3x3 weight gradient (round 2, csrc/wgrad3x3.hip)
W3_CASES = [
    # N, H, W, Cin, Cout
    (2, 5, 7, 64, 64),          # frame smaller than one 64-pixel stage, halo larger than an image
    (1, 1, 1, 64, 64),          # a single pixel: only the centre tap sees data
    (2, 20, 22, 64, 64),
    (3, 32, 32, 256, 256),      # layer-3 shape (smaller batch): 16 tiles
    (2, 63, 63, 128, 128),      # layer 2: halo of two chunks (W + 3 = 66 > 64)
    (1, 125, 125, 64, 64),      # layer 1: the widest frame the 512-row ring takes
    (2, 30, 17, 128, 192),      # rectangular, Cout not a power of two
]


@pytest.mark.parametrize("splitk", [0, 1, 3, 7])
@pytest.mark.parametrize("case", W3_CASES)
def test_wgrad3x3_all_taps(case, splitk):
    """The zero-padded-frame reduction with nine accumulator sets vs torch autograd, for several split-K factors (slice
    boundaries in the middle of rows / images), packed and OIHW outputs; and vs the per-tap kernel it replaces."""
    from tinyfaces import ops
    dtype = torch.bfloat16
    N, H, W, Cin, Cout = case
    g = _g(hash(case) % 997)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(q(x, dtype), w, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(q(gy, dtype))
    xd, gyd = to_nhwc(x, dtype), to_nhwc(gy, dtype)
    dw = ops.conv2d_wgrad(xd, gyd, Cin, Cout, 3, 3, 1, 1, tile=3, splitk=splitk)
    d = err(dw.cpu(), w.grad)
    dwp = ops.conv2d_wgrad(xd, gyd, Cin, Cout, 3, 3, 1, 1, tile=3, splitk=splitk, packed=True)     # [Cout][tap][Cin]
    dp = err(dwp.cpu().reshape(Cout, 9, Cin).permute(0, 2, 1).reshape(Cout, Cin, 3, 3), w.grad)
    old = ops.conv2d_wgrad(xd, gyd, Cin, Cout, 3, 3, 1, 1, tile=1)
    do = err(dw.cpu(), old.cpu())
    # two-phase epilogue (partial tiles + summing kernel) ACCUMULATES into dW like the atomic form: start from a non-zero buffer
    base = torch.randn(Cout, Cin, 3, 3, generator=g)
    dw2 = ops.conv2d_wgrad(xd, gyd, Cin, Cout, 3, 3, 1, 1, tile=3, splitk=splitk, out=base.clone().cuda(), two_phase=True)
    d2 = err(dw2.cpu() - base, w.grad)
    report(f"wgrad3x3[{case},sk{splitk}]", rel=d[2], packed_rel=dp[2], vs_per_tap=do[2], two_phase_rel=d2[2])
    assert d[2] < 2e-3 and dp[2] < 2e-3 and do[2] < 1e-4 and d2[2] < 2e-3


PWG_CASES = [
    # N, H, W, [(Cin, Cout, ldx, lddy)]
    (2, 9, 7, [(128, 128, 128, 128), (256, 128, 256, 128)]),                          # M = 126: a ragged last stage, several k-steps
    (1, 5, 5, [(64, 192, 64, 192), (320, 64, 320, 64), (40, 24, 40, 24)]),           # channel counts that are no multiple of the 128 tile
    (3, 11, 13, [(125, 512, 128, 512), (512, 125, 512, 128)]),                        # padded leading dimensions (the head shape)
    (1, 32, 33, [(1024, 256, 1024, 256), (256, 1024, 256, 1024)] * 2),               # the layer-3 pair, M = 1056
]


@pytest.mark.parametrize("case", PWG_CASES)
def test_wgrad_group_pointwise(case):
    """tf_conv2d_wgrad_group, pointwise form (csrc/wgrad_group.hip): every 128 x 128 tile reduced over all pixels in-block, dW OVERWRITTEN
    (the outputs start as NaN), vs an fp32 matmul of the same bf16-rounded operands."""
    from tinyfaces import ops
    dtype = torch.bfloat16
    N, H, W, probs = case
    g = _g(len(probs) * 131 + H)
    M = N * H * W
    ins, refs = [], []
    for Cin, Cout, ldx, lddy in probs:
        x = torch.zeros(M, ldx); dy = torch.zeros(M, lddy)
        x[:, :Cin] = torch.randn(M, Cin, generator=g); dy[:, :Cout] = torch.randn(M, Cout, generator=g)
        if ldx > Cin: x[:, Cin:] = 7.0                       # pad columns hold garbage: must not leak into dW
        if lddy > Cout: dy[:, Cout:] = -3.0
        refs.append(q(dy[:, :Cout], dtype).double().t() @ q(x[:, :Cin], dtype).double())
        ins.append((x.view(N, H, W, ldx).to("cuda", dtype), dy.view(N, H, W, lddy).to("cuda", dtype), Cin, Cout))
    outs = ops.conv2d_wgrad_group(ins, 1, 0)
    worst = 0.0
    for dw, ref in zip(outs, refs):
        assert torch.isfinite(dw).all()
        worst = max(worst, err(dw.cpu().reshape(ref.shape), ref)[2])
    report(f"wgrad_group_pw[{N}x{H}x{W},{len(probs)}]", rel=worst)
    assert worst < 1e-4                                      # fp32 accumulation of exact bf16 products: order only


def test_wgrad_group_pointwise_layer3_full_k():
    """The real shape: M = 12 288 pixels (384 stages per tile), three problems of each kind -- and the same gradients from the split-K
    kernel it replaces."""
    from tinyfaces import ops
    dtype = torch.bfloat16
    g = _g(77)
    N, H, W = 12, 32, 32
    ins, refs = [], []
    for Cin, Cout in [(1024, 256), (256, 1024)] * 3:
        x = torch.randn(N * H * W, Cin, generator=g); dy = torch.randn(N * H * W, Cout, generator=g) * 0.1
        refs.append(q(dy, dtype).double().t() @ q(x, dtype).double())
        ins.append((x.view(N, H, W, Cin).to("cuda", dtype), dy.view(N, H, W, Cout).to("cuda", dtype), Cin, Cout))
    outs = ops.conv2d_wgrad_group(ins, 1, 0)
    worst = max(err(dw.cpu().reshape(ref.shape), ref)[2] for dw, ref in zip(outs, refs))
    old = ops.conv2d_wgrad(ins[0][0], ins[0][1], 1024, 256, 1, 1, 1, 0)
    vs_old = err(outs[0].cpu(), old.cpu())[2]
    report("wgrad_group_pw[layer3]", rel=worst, vs_splitk=vs_old)
    assert worst < 1e-4 and vs_old < 1e-4


@pytest.mark.parametrize("case", [(2, 9, 11, 64, 64, 3), (12, 32, 32, 256, 256, 2), (1, 30, 17, 128, 192, 4)])
def test_wgrad_group_3x3(case):
    """tf_conv2d_wgrad_group, 3x3 form: n problems of one shape through the all-taps kernel with splitk = 1 and plain stores, vs torch
    autograd and vs the single-problem kernel."""
    from tinyfaces import ops
    dtype = torch.bfloat16
    N, H, W, Cin, Cout, n = case
    g = _g(sum(case))
    ins, refs = [], []
    for _ in range(n):
        x = torch.randn(N, Cin, H, W, generator=g)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
        y = F.conv2d(q(x, dtype), w, padding=1)
        gy = torch.randn(y.shape, generator=g)
        y.backward(q(gy, dtype))
        refs.append(w.grad)
        ins.append((to_nhwc(x, dtype), to_nhwc(gy, dtype), Cin, Cout))
    outs = ops.conv2d_wgrad_group(ins, 3, 1)
    worst = max(err(dw.cpu(), ref)[2] for dw, ref in zip(outs, refs))
    single = ops.conv2d_wgrad(ins[-1][0], ins[-1][1], Cin, Cout, 3, 3, 1, 1, tile=3, splitk=1)
    vs_single = err(outs[-1].cpu(), single.cpu())[2]
    report(f"wgrad_group_3x3[{case}]", rel=worst, vs_single=vs_single)
    assert worst < 2e-3 and vs_single < 1e-6


def test_wgrad_group_refuses_mixed_or_foreign_problems():
    from tinyfaces import ops
    x = torch.randn(1, 8, 8, 64, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):                       # fp32 operands: the grouped kernels are bf16 only
        ops.conv2d_wgrad_group([(x.float(), x.float(), 64, 64)], 1, 0)
    x2 = torch.randn(1, 9, 8, 64, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):                       # two pixel counts in one pointwise group
        ops.conv2d_wgrad_group([(x, x, 64, 64), (x2, x2, 64, 64)], 1, 0)
    with pytest.raises(RuntimeError):                       # two shapes in one 3x3 group
        ops.conv2d_wgrad_group([(x, x, 64, 64), (x2, x2, 64, 64)], 3, 1)


def test_wgrad3x3_refuses_what_it_cannot_do():
    from tinyfaces import ops
    x = torch.randn(1, 9, 9, 64, device="cuda").to(torch.bfloat16)
    gy = torch.randn(1, 5, 5, 64, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):                       # stride 2: the per-tap kernel's job
        ops.conv2d_wgrad(x, gy, 64, 64, 3, 3, 2, 1, tile=3)
    gy1 = torch.randn(1, 9, 9, 64, device="cuda").to(torch.bfloat16)
    with pytest.raises(RuntimeError):                       # 1x1
        ops.conv2d_wgrad(x, gy1, 64, 64, 1, 1, 1, 0, tile=3)


@pytest.mark.parametrize("ratio", [1.0, 10.0, 100.0])
def test_bn_statistics_with_large_mean(ratio):
    """Batch statistics come from fp32 (sum, sum of squares): var = E[x^2] - mean^2 cancels (mean/std)^2 of the significant bits that
    torch's Welford pass keeps.  This bounds the resulting error of invstd and of the normalised output against torch for
    |mean| = ratio x std (a bias-free conv of ReLU activations sits at ratio <~ 3 in this network): relative variance error
    <= ~8 (ratio^2 + 1) 2^-24 with fp32 partial sums.  At ratio 100 that is 5e-3 -- known and documented."""
    from tinyfaces._hip import lib, ptr, stream, tf_dtype
    dtype = torch.float32
    g = _g(int(ratio))
    N, H, W, C = 4, 24, 24, 64
    M = N * H * W
    tfd = tf_dtype(dtype)
    xr = torch.randn(N, C, H, W, generator=g) + ratio * (torch.rand(C, generator=g).view(1, C, 1, 1) + 0.5)
    gamma, beta = torch.ones(C), torch.zeros(C)
    ref = F.batch_norm(xr, None, None, gamma, beta, True, 0.1, 1e-5)
    x_d = to_nhwc(xr, dtype)
    nb = lib().tf_colstats_blocks(M, C, tfd)
    part = torch.zeros(nb, 2, C, device="cuda")
    assert lib().tf_colstats(tfd, ptr(x_d), None, ptr(x_d), None, M, C, C, ptr(part), stream()) == 0
    scale, shift, mean, invstd, rmd, rvd = [torch.zeros(C, device="cuda") for _ in range(6)]
    gd, bd = gamma.cuda(), beta.cuda()
    assert lib().tf_bn_finalize(ptr(part), nb, C, C, float(M), ptr(gd), ptr(bd), 1e-5, 0.1, ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                ptr(rmd), ptr(rvd), 0, stream()) == 0
    y = x_d.float().cpu() * scale.cpu() + shift.cpu()
    var_ref = xr.double().var(dim=(0, 2, 3), unbiased=False)
    var_rel = float(((1.0 / invstd.cpu().double() ** 2 - 1e-5 - var_ref).abs() / var_ref).max())
    d = err(from_nhwc(y.to(dtype).cuda()), ref)
    report(f"bn_large_mean[{ratio}]", var_rel=var_rel, y_maxabs=d[0])
    assert var_rel < 8 * (ratio ** 2 + 1) * 2.0 ** -24 * 4 and d[0] < 4e-6 * (ratio ** 2 + 1) + 1e-5


@pytest.mark.parametrize("ratio", [10.0, 100.0])
def test_bn_statistics_shifted_by_the_running_mean(ratio):
    """r3: the conv epilogue sums (x - s) and (x - s)^2 with s = the BN's running mean (tf_conv_args.stat_shift) and the
    consumer's table rebuilds mean = s + E[x - s], var = E[(x-s)^2] - E[x-s]^2.  With |mean| = ratio x std and a shift within one std of
    the mean the variance is right to fp32 rounding (the unshifted sums of the test above are off by 8 (ratio^2 + 1) 2^-24: 5e-3 at 100);
    with no shift the old arithmetic is unchanged.  An identity pointwise conv in fp32 (exact MFMA) feeds tf_bn_relu_fused."""
    import ctypes as C
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream
    g = _g(int(ratio) + 7)
    N, H, W, Cc = 4, 24, 24, 64
    M = N * H * W
    center = ratio * (torch.rand(Cc, generator=g) + 0.5)
    xr = torch.randn(N, Cc, H, W, generator=g) + center.view(1, Cc, 1, 1)
    x_d = to_nhwc(xr, torch.float32)
    wp = ops.pack_weight(torch.eye(Cc).view(Cc, Cc, 1, 1).cuda(), torch.float32)
    var_ref = xr.double().var(dim=(0, 2, 3), unbiased=False)
    mean_ref = xr.double().mean(dim=(0, 2, 3))
    out = {}
    for name, shift in (("none", None), ("running_mean", (center + 0.7 * torch.randn(Cc, generator=g)).cuda())):
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_F32, 0
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cc, H, W, Cc, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Cc, _hip.EPI_STATS, 0
        y = torch.empty(N, H, W, Cc, device="cuda")
        rows = lib().tf_conv_mtiles(C.byref(a))
        stat = torch.zeros(rows * 2 + 1, Cc, device="cuda")            # [rows][2][C] + the shift row
        a.x, a.w, a.y, a.stat_out = ptr(x_d), ptr(wp), ptr(y), ptr(stat)
        if shift is not None:
            a.stat_shift, a.stat_shift_out = ptr(shift), stat[rows * 2].data_ptr()
        assert lib().tf_conv2d(C.byref(a), stream()) == 0
        assert torch.equal(y, x_d)                                    # identity conv, exact fp32 MFMA
        vec = [torch.zeros(Cc, device="cuda") for _ in range(6)]
        gam, bet = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        d = _hip.BnFwdDesc()
        d.stat, d.gamma, d.beta = ptr(stat), ptr(gam), ptr(bet)
        d.scale, d.shift, d.mean, d.invstd, d.running_mean, d.running_var = [ptr(v) for v in vec]
        d.stat_shift = stat[rows * 2].data_ptr()
        z = torch.empty_like(y)
        assert lib().tf_bn_relu_fused(_hip.TF_F32, ptr(y), C.byref(d), rows, M, Cc, float(M), 1e-5, 0.1, ptr(z), stream()) == 0
        torch.cuda.synchronize()
        if shift is not None:
            assert torch.equal(stat[rows * 2], shift)                 # the producer recorded what it subtracted
        var = 1.0 / vec[3].cpu().double() ** 2 - 1e-5
        out[name] = (float(((var - var_ref).abs() / var_ref).max()), float((vec[2].cpu().double() - mean_ref).abs().max()))
        zr = torch.relu(F.batch_norm(xr, None, None, torch.ones(Cc), torch.zeros(Cc), True, 0.1, 1e-5))
        out[name + "_y"] = err(from_nhwc(z), zr)[0]
    report(f"bn_stat_shift[{ratio}]", var_rel_none=out["none"][0], var_rel_shift=out["running_mean"][0], mean_abs_none=out["none"][1],
           mean_abs_shift=out["running_mean"][1], y_none=out["none_y"], y_shift=out["running_mean_y"])
    assert out["running_mean"][0] < 2e-5 and out["running_mean_y"] < 2e-5          # fp32 rounding of the sums, whatever the ratio
    assert out["none"][0] < 8 * (ratio ** 2 + 1) * 2.0 ** -24 * 4                  # unchanged without a shift (bound of the test above)
    assert out["running_mean"][0] < out["none"][0] or out["none"][0] < 1e-5


@pytest.mark.parametrize("case", [(2, 9, 11, 256, 128), (1, 16, 16, 512, 256), (3, 7, 13, 1024, 256), (2, 8, 8, 128, 128)])
def test_conv_pwx_bn_backward_prologue(case):
    """r3, csrc/conv_pwx.hip: tf_conv2d_bnbwd == tf_bn_bwd_apply_fused followed by the pointwise data gradient (conv_dma) on the applied
    tensor, in ONE launch: the side output T1 = A*g + B*x + D (bf16), the masked gradient with its BN-backward sums (MASK | STATS2), the
    published dgamma / dbeta -- and the plain form of the kernel (tile 60, no prologue, STATS epilogue) == conv_dma.  M is not a multiple of
    the 64-pixel tile in three of the four cases."""
    from tinyfaces import _hip as _hip_x
    if not _hip_x.experimental():
        pytest.skip("conv_pwx is compiled into the experimental build only (build.py --experimental)")
    import ctypes as C
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream
    N, H, W, K, Co = case
    M = N * H * W
    g = _g(sum(case))
    dt = torch.bfloat16
    gz = torch.randn(N, H, W, K, generator=g).to(dt).cuda()
    x2 = (torch.randn(N, H, W, K, generator=g) * 1.5 + 0.3).to(dt).cuda()
    w = torch.randn(K, Co, 1, 1, generator=g) / K ** 0.5                       # forward weight [Cout_fwd = K][Cin_fwd = Co]: the data gradient maps K -> Co
    wt = ops.pack_weight(w.cuda(), dt, transpose=True)
    cprev = torch.randn(N, H, W, Co, generator=g).to(dt).cuda()                # the tensor behind the ReLU mask (c2)
    ms, mh = (torch.rand(Co, generator=g) + 0.5).cuda(), (torch.randn(Co, generator=g) * 0.2).cuda()
    rows = lib().tf_get_stat_rows()
    stat = (torch.randn(rows, 2, K, generator=g) * 3).cuda()
    gamma, mean, invstd = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda(), (torch.rand(K, generator=g) + 0.5).cuda()

    def desc(dgam, dbet):
        d = _hip.BnBwdDesc()
        d.stat, d.gamma, d.mean, d.invstd, d.dgamma, d.dbeta, d.nk, d.kidx = ptr(stat), ptr(gamma), ptr(mean), ptr(invstd), ptr(dgam), ptr(dbet), 2, 1
        return d

    def conv_args(x, y, st, tile=0):
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 1
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, K, H, W, Co, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Co, _hip.EPI_MASK | _hip.EPI_STATS2, tile
        a.x, a.w, a.y, a.aux, a.mask_scale, a.mask_shift, a.stat_out = ptr(x), ptr(wt), ptr(y), ptr(cprev), ptr(ms), ptr(mh), ptr(st)
        return a
    # reference: two launches
    t1_ref = torch.empty_like(gz)
    dg_ref, db_ref = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    d0 = desc(dg_ref, db_ref)
    assert lib().tf_bn_bwd_apply_fused(_hip.TF_BF16, ptr(gz), None, ptr(x2), C.byref(d0), rows, M, K, float(M), ptr(t1_ref), stream()) == 0
    y_ref = torch.empty(N, H, W, Co, dtype=dt, device="cuda")
    a0 = conv_args(t1_ref, y_ref, None, tile=13)
    st_ref = torch.zeros(lib().tf_conv_mtiles(C.byref(a0)), 2, Co, device="cuda")
    a0.stat_out = ptr(st_ref)
    assert lib().tf_conv2d(C.byref(a0), stream()) == 0
    # fused: one launch
    t1 = torch.zeros_like(gz)
    dg, db = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    d1 = desc(dg, db)
    y = torch.empty_like(y_ref)
    mt = min(rows, (M + 63) // 64)
    st = torch.zeros(mt, 2, Co, device="cuda")
    a1 = conv_args(gz, y, st)
    assert lib().tf_conv2d_bnbwd(C.byref(a1), C.byref(d1), ptr(x2), ptr(t1), rows, float(M), stream()) == 0
    torch.cuda.synchronize()
    d_t1 = err(t1.float().cpu(), t1_ref.float().cpu())
    d_y = err(y.float().cpu(), y_ref.float().cpu())
    s_f, s_r = st.sum(0).cpu(), st_ref.sum(0).cpu()
    d_s = float((s_f - s_r).abs().max() / (s_r.abs().max() + 1e-30))
    report(f"conv_pwx_bnbwd[{case}]", t1_rel=d_t1[2], y_rel=d_y[2], stat_rel=d_s, dgamma=float((dg - dg_ref).abs().max()), dbeta=float((db - db_ref).abs().max()))
    assert d_t1[2] < 8e-3                      # one bf16 rounding of the applied tensor (the two kernels may contract the FMAs differently)
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    assert d_y[2] < 2e-2 and d_s < 2e-2        # bf16 operands that differ by <= 1 ulp in a few places, fp32 accumulation
    # independent check of the gradient against torch on the fused kernel's own T1 (exactly representable operands)
    ref = F.conv_transpose2d(t1.float().cpu().permute(0, 3, 1, 2), q(w, dt)) if False else torch.einsum("nhwk,kc->nhwc", t1.float().cpu(), q(w, dt)[:, :, 0, 0])
    ref = torch.where(cprev.float().cpu() * ms.cpu() + mh.cpu() > 0, ref, torch.zeros_like(ref))
    assert err(y.float().cpu(), ref)[2] < 6e-3
    # plain form (tile 60, forward-style STATS epilogue) == conv_dma
    wf = ops.pack_weight(torch.randn(Co, K, 1, 1, generator=g).cuda() / K ** 0.5, dt)
    ya, sa = ops.conv2d_nhwc(gz, wf, Co, 1, 1, 1, 0, epi=_hip.EPI_STATS, want_stats=True, tile=13)
    yb, sb = ops.conv2d_nhwc(gz, wf, Co, 1, 1, 1, 0, epi=_hip.EPI_STATS, want_stats=True, tile=60)
    assert err(yb.float().cpu(), ya.float().cpu())[2] < 4e-3
    assert float((sa.sum(0) - sb.sum(0)).abs().max() / sa.sum(0).abs().max()) < 1e-3


@pytest.mark.parametrize("case", [(2, 9, 11, 64, 256, False), (1, 16, 16, 128, 512, True), (3, 7, 13, 256, 1024, True), (2, 8, 8, 64, 64, False)])
def test_conv_bn_relu_forward_prologue_in_lds(case):
    """r3, conv_dma.hip (ring-less pointwise kernel, tf_conv_args.bnf): tf_conv2d with the BatchNorm + ReLU of its input applied to the pixel
    tile in LDS == tf_bn_relu_fused followed by the same tf_conv2d, BIT for bit: the activated tensor (bnf_out), the conv output, its
    statistic rows, and everything tf_bn_relu_fused publishes (scale / shift / mean / invstd, running statistics).  M is not a multiple of
    the 128-pixel tile in two of the cases; with and without the statistic shift row."""
    from tinyfaces import _hip as _hip_x
    if not _hip_x.experimental():
        pytest.skip("the in-LDS BatchNorm prologue (tf_conv_args.bnf) is compiled into the experimental build only (build.py --experimental)")
    import ctypes as C
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream
    N, H, W, K, Co, shifted = case
    M = N * H * W
    g = _g(sum(case[:5]))
    dt = torch.bfloat16
    x = (torch.randn(N, H, W, K, generator=g) * 1.3 + 0.4).to(dt).cuda()
    w = ops.pack_weight((torch.randn(Co, K, 1, 1, generator=g) / K ** 0.5).cuda(), dt)
    rows = 5
    stat = torch.zeros(rows * 2 + 1, K, device="cuda")
    xs = x.float().view(M, K)
    sh = (torch.randn(K, generator=g) * 0.2 + 0.4).cuda() if shifted else torch.zeros(K, device="cuda")
    for r in range(rows):                                             # statistic rows of a producer that folded its tiles into 5 rows
        part = xs[r::rows] - sh
        stat[2 * r], stat[2 * r + 1] = part.sum(0), (part * part).sum(0)
    stat[rows * 2] = sh
    gam, bet = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()

    def run(fused):
        vec = [torch.zeros(K, device="cuda") for _ in range(4)] + [torch.full((K,), 0.25, device="cuda"), torch.full((K,), 2.0, device="cuda")]
        d = _hip.BnFwdDesc()
        d.stat, d.gamma, d.beta = ptr(stat), ptr(gam), ptr(bet)
        d.scale, d.shift, d.mean, d.invstd, d.running_mean, d.running_var = [ptr(v) for v in vec]
        if shifted:
            d.stat_shift = stat[rows * 2].data_ptr()
        act = torch.zeros(N, H, W, K, dtype=dt, device="cuda")
        y = torch.zeros(N, H, W, Co, dtype=dt, device="cuda")
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 0
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, K, H, W, Co, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Co, _hip.EPI_STATS, 0
        srows = lib().tf_conv_mtiles(C.byref(a))
        so = torch.zeros(srows * 2, Co, device="cuda")
        a.w, a.y, a.stat_out = ptr(w), ptr(y), ptr(so)
        if fused:
            a.x, a.bnf, a.bnf_out, a.bnf_rows, a.bnf_count, a.bnf_eps, a.bnf_momentum = ptr(x), C.addressof(d), ptr(act), rows, float(M), 1e-5, 0.1
        else:
            assert lib().tf_bn_relu_fused(_hip.TF_BF16, ptr(x), C.byref(d), rows, M, K, float(M), 1e-5, 0.1, ptr(act), stream()) == 0
            a.x = ptr(act)
        assert lib().tf_conv2d(C.byref(a), stream()) == 0
        torch.cuda.synchronize()
        return [act, y, so] + vec

    two, one = run(False), run(True)
    names = ["activated", "y", "stat_rows", "scale", "shift", "mean", "invstd", "running_mean", "running_var"]
    for nm, p, q in zip(names, two, one):
        assert torch.equal(p, q), (nm, float((p.float() - q.float()).abs().max()))
    ref = torch.relu((xs - xs.mean(0)) / torch.sqrt(xs.var(0, unbiased=False) + 1e-5) * gam + bet)
    assert err(one[0].float().view(M, K).cpu(), ref.cpu())[0] < 2e-2
    # a shape the ring-less kernel does not take says so instead of ignoring the descriptor
    a = _hip.ConvArgs()
    a.dtype, a.mode = _hip.TF_BF16, 0
    a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = 1, 8, 8, 512, 8, 8, 64, 1, 1, 1, 0
    a.ldy = 64
    xx, ww, yy = torch.zeros(64, 512, dtype=dt, device="cuda"), torch.zeros(512 * 64, dtype=dt, device="cuda"), torch.zeros(64, 64, dtype=dt, device="cuda")
    d = _hip.BnFwdDesc()
    a.x, a.w, a.y, a.bnf = ptr(xx), ptr(ww), ptr(yy), C.addressof(d)
    assert lib().tf_conv2d(C.byref(a), stream()) == -3                # TF_ERR_UNSUPPORTED


@pytest.mark.parametrize("case", [(2, 9, 11, 256, 128, False, True), (1, 16, 16, 512, 256, True, False), (3, 7, 13, 1024, 256, False, True),
                                  (2, 8, 8, 128, 128, True, True), (12, 32, 32, 1024, 256, False, True)])
def test_conv_pwx_bn_forward_prologue(case):
    """r5, csrc/conv_pwx.hip: tf_conv2d_bnfwd == tf_bn_add_relu_fused followed by the pointwise conv (conv_dma) on its output, in ONE launch:
    the side output y = relu(bn(x) + (bn_res(res) | res)) (bf16), the conv output with its statistic rows (STATS, with and without the
    shift row), and everything tf_bn_add_relu_fused publishes for BOTH BatchNorms (scale / shift / mean / invstd, running statistics).
    K = 128 has fewer stages (2) than the pixel ring is deep; M is not a multiple of the 64-pixel tile in three cases; the last case is
    the layer-3 shape of the bs = 12 step (192 blocks, 16 stages)."""
    from tinyfaces import _hip as _hip_x
    if not _hip_x.experimental():
        pytest.skip("conv_pwx is compiled into the experimental build only (build.py --experimental)")
    import ctypes as C
    from tinyfaces import _hip, ops
    from tinyfaces._hip import lib, ptr, stream
    N, H, W, K, Co, ds, shifted = case
    M = N * H * W
    g = _g(sum(case[:5]) + 3)
    dt = torch.bfloat16
    x = (torch.randn(N, H, W, K, generator=g) * 1.3 + 0.4).to(dt).cuda()
    res = (torch.randn(N, H, W, K, generator=g) * 0.9 - 0.2).to(dt).cuda()
    w0 = torch.randn(Co, K, 1, 1, generator=g) / K ** 0.5
    w = ops.pack_weight(w0.cuda(), dt)
    rows = lib().tf_get_stat_rows()

    def stat_of(t, sh):
        st = torch.zeros(rows * 2 + 1, K, device="cuda")
        ts = t.float().view(M, K)
        for r in range(rows):
            part = ts[r::rows] - sh
            st[2 * r], st[2 * r + 1] = part.sum(0), (part * part).sum(0)
        st[rows * 2] = sh
        return st
    sh1 = (torch.randn(K, generator=g) * 0.2 + 0.4).cuda() if shifted else torch.zeros(K, device="cuda")
    sh2 = (torch.randn(K, generator=g) * 0.2 - 0.2).cuda() if shifted else torch.zeros(K, device="cuda")
    stat1, stat2 = stat_of(x, sh1), stat_of(res, sh2)
    gam = [(torch.rand(K, generator=g) + 0.5).cuda() for _ in range(2)]
    bet = [(torch.randn(K, generator=g) * 0.3).cuda() for _ in range(2)]
    oshift = (torch.randn(Co, generator=g) * 0.1).cuda()            # the shift of the conv's OWN statistic sums (its BN's running mean)

    def run(fused):
        vecs = [[torch.zeros(K, device="cuda") for _ in range(4)] + [torch.full((K,), 0.25, device="cuda"), torch.full((K,), 2.0, device="cuda")] for _ in range(2)]
        descs = []
        for i, (st, vec) in enumerate(zip((stat1, stat2), vecs)):
            d = _hip.BnFwdDesc()
            d.stat, d.gamma, d.beta = ptr(st), ptr(gam[i]), ptr(bet[i])
            d.scale, d.shift, d.mean, d.invstd, d.running_mean, d.running_var = [ptr(v) for v in vec]
            if shifted:
                d.stat_shift = st[rows * 2].data_ptr()
            descs.append(d)
        yact = torch.zeros(N, H, W, K, dtype=dt, device="cuda")
        y = torch.zeros(N, H, W, Co, dtype=dt, device="cuda")
        a = _hip.ConvArgs()
        a.dtype, a.mode = _hip.TF_BF16, 0
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, K, H, W, Co, 1, 1, 1, 0
        a.ldy, a.epi, a.tile = Co, _hip.EPI_STATS, (0 if fused else 13)
        srows = min(rows, (M + 63) // 64) if fused else lib().tf_conv_mtiles(C.byref(a))
        so = torch.zeros(srows, 2, Co, device="cuda")
        so_shift = torch.zeros(Co, device="cuda")
        a.w, a.y, a.stat_out = ptr(w), ptr(y), ptr(so)
        if shifted:
            a.stat_shift, a.stat_shift_out = ptr(oshift), ptr(so_shift)
        bn_r = C.byref(descs[1]) if ds else None
        if fused:
            a.x = ptr(x)
            assert lib().tf_conv2d_bnfwd(C.byref(a), C.byref(descs[0]), ptr(res), bn_r, ptr(yact), rows, float(M), 1e-5, 0.1, stream()) == 0
        else:
            assert lib().tf_bn_add_relu_fused(_hip.TF_BF16, ptr(x), C.byref(descs[0]), ptr(res), bn_r, rows, M, K, float(M), 1e-5, 0.1, ptr(yact), stream()) == 0
            a.x = ptr(yact)
            assert lib().tf_conv2d(C.byref(a), stream()) == 0
        torch.cuda.synchronize()
        return yact, y, so.sum(0), so_shift, vecs[0], (vecs[1] if ds else [])

    two, one = run(False), run(True)
    d_act = err(one[0].float().cpu(), two[0].float().cpu())
    d_y = err(one[1].float().cpu(), two[1].float().cpu())
    d_s = float((one[2] - two[2]).abs().max() / (two[2].abs().max() + 1e-30))
    report(f"conv_pwx_bnfwd[{case}]", act_rel=d_act[2], y_rel=d_y[2], stat_rel=d_s)
    assert d_act[2] < 8e-3                       # at most one bf16 rounding apart (the two kernels may contract the FMAs differently)
    assert float((one[0] != two[0]).float().mean()) < 2e-2
    assert d_y[2] < 2e-2 and d_s < 2e-2
    assert torch.equal(one[3], two[3])
    for nm, pa, pb in zip(["scale", "shift", "mean", "invstd", "running_mean", "running_var"] * 2, two[4] + two[5], one[4] + one[5]):
        assert torch.equal(pa, pb), (nm, float((pa - pb).abs().max()))
    # independent check: the activation against torch's BatchNorm arithmetic, the conv against the kernel's own activation
    xs, rs = x.float().view(M, K), res.float().view(M, K)
    bn = lambda t, i: (t - t.mean(0)) / torch.sqrt(t.var(0, unbiased=False) + 1e-5) * gam[i] + bet[i]
    ref = torch.relu(bn(xs, 0) + (bn(rs, 1) if ds else rs))
    assert err(one[0].float().view(M, K).cpu(), ref.cpu())[0] < 4e-2
    assert err(one[1].float().view(M, Co).cpu(), one[0].float().view(M, K).cpu() @ q(w0, dt)[:, :, 0, 0].t())[2] < 6e-3


# ------------------------------------------------------------------ wave-autonomous streaming pointwise kernel (round 5, csrc/conv_pws.hip)
PWS_SHAPES = [(64, 256), (256, 64), (64, 64), (256, 128), (256, 1024), (128, 512), (256, 512)]      # the last three: sliced, 128 output channels per block


@pytest.mark.parametrize("dtype", HALF)
@pytest.mark.parametrize("shape", PWS_SHAPES)
def test_conv_pws_against_the_tiled_kernel(dtype, shape):
    """r5, csrc/conv_pws.hip (tile code 70): the persistent wave-streaming pointwise conv for the short-K / large-M launches == the tiled
    LDS-DMA kernel (tile 13) on the same operands -- same products, another summation order -- for every epilogue set it takes: plain,
    STATS with a shift (sum of the rows = the tile kernel's), AFFINE + RELU, AFFINE, AFFINE + RES + RELU, MASK + STATS2; M = 16 637 pixels
    (not a multiple of the 16-pixel wave tile: a ragged last tile; 1040 tiles over 4 x 256 waves: most waves own one tile, some none);
    and against torch on exactly representable operands.  fp16: the inference sets only."""
    from tinyfaces import _hip, ops
    Cin, Cout = shape
    if Cout > 256 and not _hip.experimental():
        pytest.skip("the output-channel slices of conv_pws are compiled into the experimental build only (build.py --experimental)")
    N, H, W = 1, 127, 131
    M = N * H * W
    g = _g(Cin * 3 + Cout)
    x = torch.randn(N, H, W, Cin, generator=g).to(dtype).cuda()
    w0 = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    wp = ops.pack_weight(w0.cuda(), dtype)
    aux = (torch.randn(N, H, W, Cout, generator=g) * 0.7).to(dtype).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.2).cuda()
    ref = torch.einsum("nhwk,ck->nhwc", x.float().cpu(), q(w0, dtype)[:, :, 0, 0])
    E = _hip
    sets = [("plain", 0, {}), ("affine_relu", E.EPI_AFFINE | E.EPI_RELU, dict(epi_scale=sc, epi_shift=sh)), ("affine", E.EPI_AFFINE, dict(epi_scale=sc, epi_shift=sh))]
    sets.append(("affine_res_relu", E.EPI_AFFINE | E.EPI_RES | E.EPI_RELU, dict(epi_scale=sc, epi_shift=sh, aux=aux)))
    if dtype == torch.float16:
        sets = [s for s in sets if s[0] != "plain"]
    worst = {}
    for name, epi, kw in sets:
        a = ops.conv2d_nhwc(x, wp, Cout, 1, 1, 1, 0, epi=epi, tile=13, **kw)
        b = ops.conv2d_nhwc(x, wp, Cout, 1, 1, 1, 0, epi=epi, tile=70, **kw)
        worst[name] = err(b.float().cpu(), a.float().cpu())[2]
        assert worst[name] < TOL_H[dtype], (name, worst[name])
        if name == "plain":
            assert err(b.float().cpu(), ref)[2] < TOL_H[dtype]
        if name == "affine_res_relu":
            want = torch.relu(ref * sc.cpu() + sh.cpu() + aux.float().cpu())
            assert err(b.float().cpu(), want)[2] < TOL_H[dtype]
    if dtype == torch.bfloat16:
        shift = (torch.randn(Cout, generator=g) * 0.1).cuda()
        # STATS with a shift: rows of the statistic buffer differ (tiles vs blocks), their sums must agree
        def stats(tile, epi, **kw):
            import ctypes as C
            a = E.ConvArgs()
            a.dtype, a.mode = E.TF_BF16, 0
            a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cin, H, W, Cout, 1, 1, 1, 0
            a.ldy, a.epi, a.tile = Cout, epi, tile
            y = torch.empty(N, H, W, Cout, dtype=dtype, device="cuda")
            rows = E.lib().tf_conv_mtiles(C.byref(a))
            st = torch.zeros(rows, 2, Cout, device="cuda")
            so = torch.zeros(Cout, device="cuda")
            a.x, a.w, a.y, a.stat_out = E.ptr(x), E.ptr(wp), E.ptr(y), E.ptr(st)
            for k_, v_ in kw.items():
                setattr(a, k_, E.ptr(v_))
            if epi == E.EPI_STATS:
                a.stat_shift_out = E.ptr(so)
            rc = E.lib().tf_conv2d(C.byref(a), E.stream())
            assert rc == 0, (tile, epi, rc)
            torch.cuda.synchronize()
            return y, st.sum(0), so, rows
        y13, s13, o13, r13 = stats(13, E.EPI_STATS, stat_shift=shift)
        y70, s70, o70, r70 = stats(70, E.EPI_STATS, stat_shift=shift)
        assert r70 == E.lib().tf_get_stat_rows() and torch.equal(o13, o70) and torch.equal(o70, shift)
        worst["stats_y"] = err(y70.float().cpu(), y13.float().cpu())[2]
        worst["stats_sum"] = float((s70 - s13).abs().max() / s13.abs().max())
        assert worst["stats_y"] < TOL_H[dtype] and worst["stats_sum"] < 2e-3
        ms, mh = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.2).cuda()
        y13, s13, _, _ = stats(13, E.EPI_MASK | E.EPI_STATS2, aux=aux, mask_scale=ms, mask_shift=mh)
        y70, s70, _, _ = stats(70, E.EPI_MASK | E.EPI_STATS2, aux=aux, mask_scale=ms, mask_shift=mh)
        worst["mask_y"] = err(y70.float().cpu(), y13.float().cpu())[2]
        worst["mask_sum"] = float((s70 - s13).abs().max() / s13.abs().max())
        # (a mask decision on a value within rounding of zero may differ between the two summation orders: a handful of elements)
        assert float((y70 != y13).float().mean()) < 0.2 and worst["mask_sum"] < 5e-3
        want = torch.where(aux.float().cpu() * ms.cpu() + mh.cpu() > 0, ref, torch.zeros_like(ref))
        assert err(y70.float().cpu(), want)[2] < TOL_H[dtype]
        if Cin == 64 or Cout >= 128:
            # the hand-over sets of conv1's data gradient: + residual gradient, masked by ReLU of the previous block's output, BN3-backward sums with its c3
            yprev = torch.randn(N, H, W, Cout, generator=g).to(dtype).cuda()
            c3 = (torch.randn(N, H, W, Cout, generator=g) * 1.2).to(dtype).cuda()
            full = E.EPI_RES | E.EPI_MASK2 | E.EPI_STATS3
            y13, s13, _, _ = stats(13, full, aux=aux, aux2=yprev, aux3=c3)
            y70, s70, _, _ = stats(70, full, aux=aux, aux2=yprev, aux3=c3)
            worst["handover_y"] = err(y70.float().cpu(), y13.float().cpu())[2]
            worst["handover_sum"] = float((s70 - s13).abs().max() / s13.abs().max())
            assert worst["handover_y"] < TOL_H[dtype] and worst["handover_sum"] < 2e-3
            want = torch.where(yprev.float().cpu() > 0, ref + aux.float().cpu(), torch.zeros_like(ref))
            assert err(y70.float().cpu(), want)[2] < TOL_H[dtype]
            for epi, kw in ((E.EPI_RES | E.EPI_MASK2, dict(aux=aux, aux2=yprev)), (E.EPI_RES, dict(aux=aux))):
                a_ = ops.conv2d_nhwc(x, wp, Cout, 1, 1, 1, 0, epi=epi, tile=13, **kw)
                b_ = ops.conv2d_nhwc(x, wp, Cout, 1, 1, 1, 0, epi=epi, tile=70, **kw)
                assert err(b_.float().cpu(), a_.float().cpu())[2] < TOL_H[dtype], epi
    report(f"conv_pws[{dtype},{shape}]", **worst)


def test_conv_pws_is_what_the_dispatcher_picks_for_the_layer1_streams():
    """tile = 0 (auto): the short-K / large-M pointwise launches go to the streaming kernel, everything else stays where it was; a shape or
    epilogue it does not take is refused when asked for explicitly."""
    from tinyfaces import _hip, ops
    x = torch.randn(1, 130, 130, 64, device="cuda").to(torch.bfloat16)
    wp = ops.pack_weight(torch.randn(256, 64, 1, 1, device="cuda") / 8, torch.bfloat16)
    _hip.lib().tf_profile_enable(1)
    ops.conv2d_nhwc(x, wp, 256, 1, 1, 1, 0)                                           # M = 16 900, 64 -> 256: streams
    ops.conv2d_nhwc(x[:, :100], wp, 256, 1, 1, 1, 0)                                  # M = 13 000 < 16 384: tiled
    torch.cuda.synchronize()
    _hip.lib().tf_profile_enable(0)
    import ctypes as C
    rows = (C.c_double * (24 * 6))()
    n = _hip.lib().tf_profile_collect(rows, 24)
    kinds = {int(rows[i * 6]): int(rows[i * 6 + 1]) for i in range(n)}
    assert kinds.get(23) == 1 and kinds.get(13) == 1, kinds
    # the sliced form (N > 256) is parity-tested on request and never the dispatcher's own choice (it is slower: DESIGN.md section 7 row 54)
    x256 = torch.randn(1, 130, 130, 256, device="cuda").to(torch.bfloat16)
    w1024 = ops.pack_weight(torch.randn(1024, 256, 1, 1, device="cuda") / 16, torch.bfloat16)
    _hip.lib().tf_profile_enable(1)
    ops.conv2d_nhwc(x256, w1024, 1024, 1, 1, 1, 0)
    torch.cuda.synchronize()
    _hip.lib().tf_profile_enable(0)
    n = _hip.lib().tf_profile_collect(rows, 24)
    assert {int(rows[i * 6]) for i in range(n)} == {13}
    w512 = ops.pack_weight(torch.randn(512, 64, 1, 1, device="cuda") / 8, torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, w512, 512, 1, 1, 1, 0, tile=70)                            # 64 -> 512: not one of its shapes
