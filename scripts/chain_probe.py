"""r6: the price of an edge in a chain of dependent kernels with and without the AQL barrier bit (csrc/probe.hip kind 13).
mode 0 plain launches (kernel boundary)  1 + in-kernel completion-counter wait  2 hipExtAnyOrderLaunch + in-kernel wait  3 two streams, no events + in-kernel wait."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces._hip import lib, stream
import ctypes as C
l = lib()
l.tf_debug_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
N = 400
names = ["kernel boundary (plain launches)", "boundary + completion-counter wait", "hipExtAnyOrderLaunch + counter wait", "two streams, no events + counter wait"]
for blocks in (256, 768, 1536):
    for mode in (0, 1, 2, 3):
        def run():
            rc = l.tf_debug_probe(13, blocks, mode, buf.data_ptr(), buf.numel(), N, stream()); assert rc == 0, rc
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize()
        ctl = buf[(1 << 20) - 64:(1 << 20) - 52].view(torch.int32).tolist()
        print(f"blocks {blocks:5d} mode {mode} {names[mode]:40s}: {a.elapsed_time(b) * 1e3 / N:6.2f} us per kernel | gave up {ctl[0]} stale {ctl[1]} predecessor already complete {ctl[2]} of {blocks * (N - 1)}", flush=True)
