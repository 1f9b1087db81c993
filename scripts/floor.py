"""What does a kernel launch cost in a dependent chain on this GPU?  Empty kernels (csrc/probe.hip kind 0 with 0 iterations) of the grid /
LDS sizes the executor launches, 200 back to back on one stream: wall time per launch (HIP events around the chain); run it under
`rocprofv3 --kernel-trace --stats` for the begin -> end duration the profiler attributes to each."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
lib = _hip.lib()
buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
N = 400
print("# empty-kernel chain: us per launch (200 dependent launches on one stream)")
print("blocks | lds KiB | us/launch")
for blocks in (1, 256, 768, 3072):
    for lds in (0, 32, 128):
        def run(n):
            lib.tf_debug_probe_chain(0, blocks, lds << 10, buf.data_ptr(), buf.numel(), 0, n, _hip.stream())
        run(10); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(N); b.record(); b.synchronize()
        print(f"{blocks:6d} | {lds:7d} | {a.elapsed_time(b) / N * 1e3:6.2f}", flush=True)
# a chain that alternates an empty 768-block kernel with a 1-iteration L2 probe (a kernel that does one dependent load round)
for blocks in (768,):
    def run2(n):
        lib.tf_debug_probe_chain(1, blocks, 0, buf.data_ptr(), buf.numel(), 1, n, _hip.stream())
    run2(10); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run2(N); b.record(); b.synchronize()
    print(f"one-load-round kernel, {blocks} blocks: {a.elapsed_time(b) / N * 1e3:6.2f} us/launch")
