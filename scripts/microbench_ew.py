"""GPU micro-benchmark of the BN elementwise kernels on the layer3 shape (12288 x 1024 bf16): same buffers vs rotating
through distinct buffers (what the training forward does: every block writes fresh activations)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
from tinyfaces._hip import lib, ptr, stream

M, C = 12288, 1024
dt = torch.bfloat16
NSET = int(os.environ.get("NSET", "24"))
xs = [torch.randn(M, C, device="cuda").to(dt) for _ in range(NSET)]
rs = [torch.randn(M, C, device="cuda").to(dt) for _ in range(NSET)]
ys = [torch.empty(M, C, device="cuda", dtype=dt) for _ in range(NSET)]
s1 = torch.rand(C, device="cuda"); h1 = torch.rand(C, device="cuda")

def run(fn, rotate, reps=48):
    for i in range(4): fn(i % NSET if rotate else 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i % NSET if rotate else 0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def add_relu(i): lib().tf_bn_add_relu(_hip.TF_BF16, ptr(xs[i]), ptr(s1), ptr(h1), ptr(rs[i]), 0, 0, M, C, ptr(ys[i]), stream())
def add_relu2(i): lib().tf_bn_add_relu(_hip.TF_BF16, ptr(xs[i]), ptr(s1), ptr(h1), ptr(rs[i]), ptr(s1), ptr(h1), M, C, ptr(ys[i]), stream())
def relu(i): lib().tf_bn_relu(_hip.TF_BF16, ptr(xs[i]), ptr(s1), ptr(h1), M, C, ptr(ys[i]), stream())
def bwd_apply(i): lib().tf_bn_bwd_apply(_hip.TF_BF16, ptr(xs[i]), 0, ptr(rs[i]), ptr(s1), ptr(h1), ptr(s1), M, C, ptr(ys[i]), stream())
def bwd_apply_y(i): lib().tf_bn_bwd_apply(_hip.TF_BF16, ptr(xs[i]), ptr(ys[(i + 1) % NSET]), ptr(rs[i]), ptr(s1), ptr(h1), ptr(s1), M, C, ptr(ys[i]), stream())
def torch_add(i): torch.add(xs[i], rs[i], out=ys[i])
def torch_copy(i): ys[i].copy_(xs[i])

mb = M * C * 2 / 1e6
for name, fn, nt in (("bn_add_relu", add_relu, 3), ("bn_add_relu(ds)", add_relu2, 3), ("bn_relu", relu, 2), ("bn_bwd_apply", bwd_apply, 3),
                     ("bn_bwd_apply+y", bwd_apply_y, 4), ("torch.add", torch_add, 3), ("torch.copy", torch_copy, 2)):
    for rot in (False, True):
        us = run(fn, rot)
        print(f"{name:16s} rotate={int(rot)}  {us:7.1f} us   {nt * mb / us / 1e3 * 1e3 / 1e3:6.2f} TB/s", flush=True)
