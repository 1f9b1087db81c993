"""Stage-by-stage timeline of the halo-resident 3x3 kernel (csrc/conv3x3h.hip) on the layer-3 shape: s_memtime stamps of every
wave of the first 8 blocks at the five points of each K stage (top, after the counted vmcnt wait, after the barrier, after the DMA
issue, after the MFMAs were issued), from the instrumented instantiation (tf_debug_conv3x3h_trace).  Prints per-phase means in
shader cycles, the clock the block ran at (s_memtime vs the 100 MHz s_memrealtime), and the first stages of one wave."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops

N, H, W, C, CO = [int(v) for v in os.environ.get("SHAPE", "12,32,32,256,256").split(",")]
dt = torch.bfloat16
x = torch.randn(N, H, W, C, device="cuda").to(dt)
w = torch.randn(CO, C, 3, 3, device="cuda") * 0.05
wp = ops.pack_weight(w, dt)
for _ in range(5):
    ops.conv2d_nhwc(x, wp, CO, 3, 3, 1, 1, tile=50)
torch.cuda.synchronize()
buf = torch.zeros(8 * 8 * 64 * 8, dtype=torch.int64, device="cuda")
_hip.lib().tf_debug_conv3x3h_trace(buf.data_ptr())
for _ in range(3):
    ops.conv2d_nhwc(x, wp, CO, 3, 3, 1, 1, tile=50)
torch.cuda.synchronize()
_hip.lib().tf_debug_conv3x3h_trace(None)
t = buf.cpu().numpy().reshape(8, 8, 64, 8).astype(np.float64)
nst = 9 * (C // 64)
names = ["wait(vmcnt)", "barrier", "dma issue", "ds_read+mfma", "loop tail"]
for b in range(8):
    meta = t[b, :, 63, :]
    cyc = meta[:, 2] - meta[:, 0]; real = meta[:, 3] - meta[:, 1]
    ghz = cyc.mean() / (real.mean() * 10e-9) / 1e9
    ep = t[b, :, 62, :5]
    st = t[b, :, :nst, :5]
    pro = (st[:, 0, 0] - meta[:, 0]).mean()
    epi = " ".join(f"{nm} {v:6.0f}" for nm, v in zip(["stage-tile", "stores", "stats", "drain"], np.diff(ep, axis=1).mean(axis=0)))
    d = np.diff(st, axis=2)                                   # [wave][stage][4]: wait, barrier, issue, compute
    tail = st[:, 1:, 0] - st[:, :-1, 4]
    per_stage = (st[:, 1:, 0] - st[:, :-1, 0]).mean()
    print(f"block {b} (hw block {int(meta[0, 4])}): K loop {cyc.mean():8.0f} cycles = {real.mean() * 10e-3:6.2f} us at {ghz:4.2f} GHz | "
          f"{per_stage:6.0f} cycles/stage | " + " ".join(f"{nm} {d[:, 1:, i].mean():6.0f}" for i, nm in enumerate(names[:4])) + f" {names[4]} {tail.mean():5.0f} | prologue {pro:6.0f} | epilogue: {epi}")
b = 0
print("block 0, per wave (mean cycles per stage: wait, barrier, issue, compute):")
st = t[b, :, :nst, :5]
d = np.diff(st, axis=2)
for wv in range(8):
    print(f"  wave {wv}: " + " ".join(f"{d[wv, 1:, i].mean():6.0f}" for i in range(4)))
print("block 0 wave 0, stages 0..12 (wait, barrier, issue, compute):")
for s in range(min(13, nst)):
    print(f"  st {s:2d}: " + " ".join(f"{d[0, s, i]:6.0f}" for i in range(4)))
