"""r5: operand-streaming probe (csrc/probe.hip kind 9).  A block of a pixel-stationary GEMM owns 64 rows of an [M][1024] bf16 matrix and
pulls them through LDS by LDS-DMA, 8 KiB per stage.  Does the ADDRESS PATTERN of a stage bound the HBM rate?  seg = contiguous bytes a
stage takes from one row (128 = the GEMM's k step over 64 rows ... 2048 = four whole rows); depth = stages in flight per block.
Matrix sizes: 25 MB (192 blocks: layer 3 at bs = 12; fits the 256 MB MALL) and 400 MB (3072 blocks, streamed in 4 launches over different windows)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces._hip import lib, ptr, stream
import ctypes as C
l = lib()
l.tf_debug_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").zero_()
def run(blocks, seg, depth, windows):
    nbytes = blocks * 64 * 2048
    def once():
        for w in range(windows):
            rc = l.tf_debug_probe(9, blocks, 0, big.data_ptr() + w * nbytes, nbytes, seg | (depth << 16), stream())
            assert rc == 0, rc
    once(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    a.record()
    for _ in range(R): once()
    b.record(); b.synchronize()
    us = a.elapsed_time(b) / R / windows * 1e3
    return us, nbytes / us / 1e6
for blocks, windows, tag in ((192, 1, "25 MB resident"), (192, 8, "25 MB x 8 windows"), (768, 8, "100 MB x 8 windows"), (3072, 2, "400 MB x 2 windows")):
    for depth in (2, 4, 8):
        row = []
        for seg in (128, 256, 512, 2048):
            us, tbs = run(blocks, seg, depth, windows)
            row.append(f"seg {seg:4d}: {us:6.1f} us {tbs:5.2f} TB/s")
        print(f"{tag:20s} blocks {blocks:4d} depth {depth}: " + " | ".join(row), flush=True)
# kind 10: the hand-over data gradient's EPILOGUE alone (tile-shaped loads of 1-3 matrices + one tile-shaped store, 1536 blocks at M = 12 288)
for nin in (1, 2, 3):
    blocks = 96 * 16
    mat = 12288 * 2048
    def once(w=0):
        rc = l.tf_debug_probe(10, blocks, 0, big.data_ptr() + w * 4 * mat, (nin + 1) * mat, nin, stream()); assert rc == 0, rc
    for rep, windows in (("same buffers", 1), ("8 rotating windows", 8)):
        once(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(40): once(i % windows)
        b.record(); b.synchronize()
        us = a.elapsed_time(b) / 40 * 1e3
        print(f"tile epilogue probe: {nin} in + 1 out x 25 MB, 1536 blocks, {rep}: {us:6.1f} us  {(nin + 1) * mat / us / 1e6:5.2f} TB/s", flush=True)
# kind 11: memory skeleton of a wave-autonomous streaming pointwise conv (persistent blocks, every wave its own 16-pixel tiles, private 2-slot ring + staging)
for shape, tag, inb, outb in ((0, "64 -> 256 channels", 128, 512), (1, "256 -> 64 channels", 512, 128)):
    for px, where in ((187500 // 16 * 16, "M = 187 500 (layer 1, bs = 12)"), (307200, "M = 307 200 (1920 x 2560 level)")):
        for nw in (4, 8):
            for blocks in (256, 512, 768, 1024):
                if nw == 8 and blocks > 512: continue
                def once(w=0):
                    rc = l.tf_debug_probe(11, blocks, nw, big.data_ptr() + w * (px * 640), px * 640, px | (shape << 28), stream()); assert rc == 0, rc
                nwin = max(1, min(4, (1 << 30) // (px * 640)))
                once(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(20): once(i % nwin)
                b.record(); b.synchronize()
                us = a.elapsed_time(b) / 20 * 1e3
                print(f"wave-stream probe {tag}, {where}: {nw} waves x {blocks:4d} blocks: {us:6.1f} us  {px * (inb + outb) / us / 1e6:5.2f} TB/s", flush=True)
