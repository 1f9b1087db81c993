"""r6: does LDS-DMA traffic slow the matrix pipe?  (csrc/probe.hip kind 12.)  One 768-thread block per CU: waves 0-7 issue only MFMAs (8 per round and wave
= one conv3x3h K stage of matrix-pipe time), waves 8-11 stream `dma` 1 KiB LDS-DMA instructions per round each from an L2-resident window.  No barriers.
mode bits: 1 MFMA waves, 2 loader waves, 4 MFMA waves also read 8 fragments per round from LDS, 8 loaders use global_load -> VGPR, 16 loaders ds_write."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces._hip import lib, stream
import ctypes as C
l = lib()
l.tf_debug_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda").zero_()
ROUNDS = 2000
DMAS = [int(v) for v in os.environ.get("DMAS", "5,3,8").split(",")]
MODES = [int(v) for v in os.environ.get("MODES", "").split(",") if v]
def run(blocks, mode, dma):
    def once():
        rc = l.tf_debug_probe(12, blocks, dma, big.data_ptr(), big.numel(), (ROUNDS << 8) | mode, stream()); assert rc == 0, rc
    once(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): once()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / 3 * 1e3 / ROUNDS     # us per round
names = {1: "MFMA only", 2: "LDS-DMA only", 3: "MFMA + LDS-DMA", 5: "MFMA + fragment reads", 7: "MFMA + reads + LDS-DMA", 10: "global_load only", 11: "MFMA + global_load",
         15: "MFMA + reads + global_load", 18: "ds_write only", 19: "MFMA + ds_write", 23: "MFMA + reads + ds_write",
         35: "MFMA + LDS-DMA, loaders prio 3", 39: "MFMA + reads + LDS-DMA, prio 3", 67: "MFMA + LDS-DMA, loaders oldest", 99: "MFMA + LDS-DMA, oldest + prio 3",
         103: "MFMA + reads + LDS-DMA, oldest+prio3", 43: "MFMA + global_load, prio 3",
         129: "MFMA waves issue the DMA themselves, interleaved (dma x 8 KiB per round)", 133: "same + fragment reads"}
for blocks in (192, 256):
    for dma in DMAS:
        for mode in (MODES if MODES else (1, 2, 3, 5, 7, 10, 11, 15, 18, 19, 23)):
            if dma != 5 and mode in (1, 5): continue
            us = run(blocks, mode, dma)
            kib = 8 * dma if mode & 128 else 4 * dma
            print(f"blocks {blocks} dma/round {dma} ({kib:2d} KiB per CU and round) mode {mode:2d} {names[mode]:28s}: {us * 1e3:7.1f} ns per round"
                  + (f"  = {kib * 1024 / (us * 1e-6) / 1e9:6.1f} GB/s per CU" if mode & 130 else ""), flush=True)
