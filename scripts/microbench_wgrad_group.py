"""r4 micro-benchmark: the grouped full-K weight gradients of layer 3 (tf_conv2d_wgrad_group) against the per-problem split-K kernels
they replace, on the real shapes (bs = 12: M = 12 288 pixels).  Every problem owns its tensors (a group of eight is 0.6 GB of
operands: no cache can hold it), times are HIP-event averages over back-to-back launches; run under the tracer for true durations
(`bash scripts/gpu_job.sh ubench scripts/microbench_wgrad_group.py`)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops  # noqa: E402


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


N, H, W = 12, 32, 32
dt = torch.bfloat16
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def prob(cin, cout):
    return (torch.randn(N, H, W, cin, device="cuda").to(dt), (torch.randn(N, H, W, cout, device="cuda") * 0.1).to(dt), cin, cout)


pw = [prob(1024, 256) for _ in range(nblk)] + [prob(256, 1024) for _ in range(nblk)]
c3 = [prob(256, 256) for _ in range(nblk)]
gf_pw = sum(2.0 * N * H * W * ci * co for _, _, ci, co in pw) / 1e9
gf_c3 = sum(2.0 * N * H * W * ci * co * 9 for _, _, ci, co in c3) / 1e9
print("TINYFACES_WGRADG_NS =", os.environ.get("TINYFACES_WGRADG_NS", "default"), flush=True)      # ring depth of the grouped pointwise kernel: read once per process
us = timeit(lambda: ops.conv2d_wgrad_group(pw, 1, 0))
print(f"grouped pointwise  n={len(pw):2d}  {us:8.1f} us  {gf_pw / us * 1e3:7.1f} TFLOP/s  ({gf_pw:.0f} GFLOP)", flush=True)
us = timeit(lambda: ops.conv2d_wgrad_group(c3, 3, 1))
print(f"grouped 3x3        n={len(c3):2d}  {us:8.1f} us  {gf_c3 / us * 1e3:7.1f} TFLOP/s  ({gf_c3:.0f} GFLOP)", flush=True)
# the per-problem kernels over the same problems (split-K + atomics / two-phase), back to back on one stream
outs = [torch.zeros(co, ci, 1, 1, device="cuda") for _, _, ci, co in pw]
us = timeit(lambda: [ops.conv2d_wgrad(x, dy, ci, co, 1, 1, 1, 0, out=o) for (x, dy, ci, co), o in zip(pw, outs)])
print(f"per-problem pointwise (wgrad_dma, split-K)  {us:8.1f} us  {gf_pw / us * 1e3:7.1f} TFLOP/s  = {us / len(pw):.1f} us each", flush=True)
outs3 = [torch.zeros(co, ci, 3, 3, device="cuda") for _, _, ci, co in c3]
us = timeit(lambda: [ops.conv2d_wgrad(x, dy, ci, co, 3, 3, 1, 1, out=o, two_phase=True) for (x, dy, ci, co), o in zip(c3, outs3)])
print(f"per-problem 3x3 (wgrad3x3 two-phase)        {us:8.1f} us  {gf_c3 / us * 1e3:7.1f} TFLOP/s  = {us / len(c3):.1f} us each", flush=True)
