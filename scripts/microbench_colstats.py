"""tf_colstats (three sums over three M x C tensors: the BN-backward sums of the downsample blocks) and the head's one-tensor form, alone: us, GB/s."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
from tinyfaces._hip import lib
from tinyfaces.ops import ptr, stream
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
dt = torch.bfloat16; tfd = _hip.tf_dtype(dt)
for M, C, nk in ((187500, 256, 3), (47628, 512, 3), (12288, 1024, 3), (47628, 128, 1), (12288, 128, 1)):
    ts = [torch.randn(M, C, device="cuda").to(dt) for _ in range(nk)]
    part = torch.zeros(16 * 3 * 1024 + 1100 * 3 * 1024, device="cuda")
    args = [ptr(ts[0]), None, ptr(ts[1]) if nk > 1 else None, ptr(ts[2]) if nk > 2 else None]
    us = timeit(lambda: lib().tf_colstats(tfd, args[0], args[1], args[2], args[3], M, C, C, ptr(part), stream()))
    by = nk * M * C * 2
    print(f"M={M} C={C} nk={nk}: {us:7.1f} us  {by / 1e6:6.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)
