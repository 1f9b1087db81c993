"""conv1 of the trunk at the benchmarked size (bs = 12, 500x500): tf_stem_conv (r4, straight from the image) against tf_stem_im2col + the
pointwise GEMM of rounds 1-3.  Run under the kernel tracer for TRUE durations: scripts/gpu_job.sh ubench scripts/microbench_stem.py (PAT=stem|conv_dma)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip, ops
from tinyfaces._hip import lib, ptr, stream, tf_dtype
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (12, 500, 500)))
dtype = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 3, H, W, generator=g).cuda()
w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).cuda()
OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
col = torch.empty(N * OH * OW, 192, dtype=dtype, device="cuda")
wp = ops.pack_weight(w.reshape(64, 147, 1, 1), dtype, cols_pad=192)
for _ in range(10):
    y, st = ops.stem_conv(x, w, dtype, epi=_hip.EPI_STATS)
    assert lib().tf_stem_im2col(ptr(x), N, H, W, tf_dtype(dtype), ptr(col), 192, stream()) == 0
    y_old = ops.conv2d_nhwc(col.view(1, 1, N * OH * OW, 192), wp, 64, 1, 1, 1, 0, epi=_hip.EPI_STATS, want_stats=True)
    dw = ops.stem_wgrad(x, y)                          # (the gradient operand: any [N*OH*OW][64] tensor of the operand type)
    dw_old = ops.conv2d_wgrad(col.view(1, 1, N * OH * OW, 192), y.view(1, 1, N * OH * OW, 64), 192, 64, 1, 1, 1, 0) if hasattr(ops, "conv2d_wgrad") else None
torch.cuda.synchronize()
print("ok", float((y.float() - y_old[0].float().reshape(y.shape)).abs().max()))
