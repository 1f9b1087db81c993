"""Which pairs of HIP streams of one process really run side by side?  (r3)
NS streams, each warmed with one launch (so each owns its hardware queue, in creation order); for every pair (a, b) the same two chains
of pointwise convs (the layer-3 shape of the bs=12 step, 40 launches each) are enqueued on a and on b and timed together.  A pair that
shares a command-processor pipe takes about as long as the two chains back to back; a pair on different pipes overlaps."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
NS = int(os.environ.get("NS", "9"))
dt = torch.bfloat16
x = torch.randn(12, 32, 32, 1024, device="cuda").to(dt)
w = ops.pack_weight(torch.randn(256, 1024, 1, 1, device="cuda") * 0.03, dt)
x2 = torch.randn(12, 32, 32, 1024, device="cuda").to(dt)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NS - 1)]
def chain(s, xx, n=40):
    with torch.cuda.stream(s):
        for _ in range(n):
            ops.conv2d_nhwc(xx, w, 256, 1, 1, 1, 0)
for s in streams:
    chain(s, x, 2)
torch.cuda.synchronize()
def timed(fn, rep=3):
    best = 1e9
    for _ in range(rep):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
solo = timed(lambda: chain(streams[0], x))
print(f"one chain alone on stream 0: {solo:.2f} ms;  two chains on ONE stream: {timed(lambda: (chain(streams[0], x), chain(streams[0], x2))):.2f} ms", flush=True)
print("pair (a,b): ms for both chains  [ratio to one chain]")
for a in range(min(NS, 3)):
    line = f"a={a}: "
    for b in range(NS):
        if a == b: line += "   --   "; continue
        t = timed(lambda: (chain(streams[a], x), chain(streams[b], x2)))
        line += f" b={b}:{t:5.2f}[{t / solo:3.1f}]"
    print(line, flush=True)
