"""r5: the wave-streaming pointwise kernel (tile code 70, csrc/conv_pws.hip) beside the tiled LDS-DMA kernel (tile codes 32 / 46 / 13: what the
dispatcher picked before) on the layer shapes it takes, each with the epilogue set the executor gives that launch; one launch at a time on an
idle GPU (bash scripts/gpu_job.sh and `python scripts/microbench_pws.py`)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops, _hip as E


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def runner(dt, N, H, W, Cin, Cout, epi, tile):
    x = torch.randn(N, H, W, Cin, device="cuda").to(dt)
    wp = ops.pack_weight(torch.randn(Cout, Cin, 1, 1, device="cuda") / Cin ** 0.5, dt)
    y = torch.empty(N, H, W, Cout, dtype=dt, device="cuda")
    aux, aux2, aux3 = (torch.randn(N, H, W, Cout, device="cuda").to(dt) for _ in range(3))
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    a = E.ConvArgs()
    a.dtype, a.mode = (E.TF_BF16 if dt == torch.bfloat16 else E.TF_F16), 0
    a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cin, H, W, Cout, 1, 1, 1, 0
    a.ldy, a.epi, a.tile = Cout, epi, tile
    rows = E.lib().tf_conv_mtiles(C.byref(a))
    st = torch.zeros(max(rows, 1), 2, Cout, device="cuda")
    a.x, a.w, a.y, a.stat_out = E.ptr(x), E.ptr(wp), E.ptr(y), E.ptr(st)
    a.aux, a.aux2, a.aux3 = E.ptr(aux), E.ptr(aux2), E.ptr(aux3)
    a.epi_scale, a.epi_shift, a.mask_scale, a.mask_shift = E.ptr(sc), E.ptr(sh), E.ptr(sc), E.ptr(sh)
    keep = (x, wp, y, aux, aux2, aux3, sc, sh, st)
    def go():
        rc = E.lib().tf_conv2d(C.byref(a), E.stream())
        if rc != 0:
            raise RuntimeError(rc)
    go.keep = keep
    return go


HAND = E.EPI_RES | E.EPI_MASK2 | E.EPI_STATS3
bf, hf = torch.bfloat16, torch.float16
CASES = [  # name, dtype, N, H, W, Cin, Cout, epilogue, operand tensors of M x Cout beside y
    ("l1.conv3 fwd stats       ", bf, 12, 125, 125, 64, 256, E.EPI_STATS, 0),
    ("l1.conv1 dgrad hand-over ", bf, 12, 125, 125, 64, 256, HAND, 3),
    ("l1.conv1 fwd stats       ", bf, 12, 125, 125, 256, 64, E.EPI_STATS, 0),
    ("l1.conv3 dgrad mask+st2  ", bf, 12, 125, 125, 256, 64, E.EPI_MASK | E.EPI_STATS2, 1),
    ("l2.conv3 fwd stats       ", bf, 12, 63, 63, 128, 512, E.EPI_STATS, 0),
    ("l2.conv1 dgrad hand-over ", bf, 12, 63, 63, 128, 512, HAND, 3),
    ("l3.conv3 fwd stats       ", bf, 12, 32, 32, 256, 1024, E.EPI_STATS, 0),
    ("l3.conv1 dgrad hand-over ", bf, 12, 32, 32, 256, 1024, HAND, 3),
    ("eval 1920x2560 l1.conv3  ", hf, 1, 480, 640, 64, 256, E.EPI_AFFINE | E.EPI_RES | E.EPI_RELU, 1),
    ("eval 1920x2560 l2.conv3  ", hf, 1, 240, 320, 128, 512, E.EPI_AFFINE | E.EPI_RES | E.EPI_RELU, 1),
    ("eval 1920x2560 l3.conv3  ", hf, 1, 120, 160, 256, 1024, E.EPI_AFFINE | E.EPI_RES | E.EPI_RELU, 1),
    ("eval 960x1280 l3.conv3   ", hf, 1, 60, 80, 256, 1024, E.EPI_AFFINE | E.EPI_RES | E.EPI_RELU, 1),
]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, dt, N, H, W, Cin, Cout, epi, naux in CASES:
    if only and only not in name:
        continue
    M = N * H * W
    mb = (M * Cin + Cin * Cout + M * Cout * (1 + naux)) * 2 / 1e6
    line = f"{name} M={M:6d} {Cin:4d}->{Cout:4d} {mb:6.1f} MB |"
    for tile in (70, 32, 46, 13):
        try:
            us = timeit(runner(dt, N, H, W, Cin, Cout, epi, tile))
            line += f" t{tile}: {us:6.1f} us {mb / us:5.2f} TB/s |"
        except RuntimeError as e:
            line += f" t{tile}: refused ({e}) |"
    print(line, flush=True)
