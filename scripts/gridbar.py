"""What would a persistent ("cooperative") conv -> BN -> conv kernel pay per dependency?  A device-wide barrier inside one launch (atomic
arrival counter, agent-scope release / acquire fences: csrc/probe.hip kind 7) against the kernel boundary it would replace (a chain of empty
kernels of the same grid, kind 0).  VERDICT r3 item 2: the forward chain of a bottleneck is 6 dependent launches; a persistent kernel turns
each boundary into one of these barriers."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import _hip
lib = _hip.lib()
buf = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, n):
    fn(2); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(n); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3


print("blocks | kernel boundary (empty 256-thread kernels, us/launch) | one-counter barrier: no data, + 16 B/thread written before and read after (us) | two-level barrier: same (us)")
for blocks in (256, 512, 768, 1024):
    N = 400
    t_chain = timed(lambda n: _hip.check(lib.tf_debug_probe_chain(0, blocks, 0, buf.data_ptr(), 4096, 0, n, _hip.stream()), "chain"), N) / N
    res, bad = [], 0
    for kind in (7, 8):
        for window in (4096, buf.numel()):
            run = lambda n: _hip.check(lib.tf_debug_probe(kind, blocks, 0, buf.data_ptr(), window, n, _hip.stream()), "bar")
            t = timed(run, N)
            torch.cuda.synchronize()
            flags = buf[:16].view(torch.int32).tolist()           # [arrivals, a barrier that gave up, stale reads after a barrier]
            bad += flags[1] + flags[2]
            res.append((t - timed(run, 0)) / N)
    print(f"{blocks:6d} | {t_chain:6.2f} | {res[0]:6.2f} {res[1]:6.2f} | {res[2]:6.2f} {res[3]:6.2f} | errors {bad}", flush=True)
