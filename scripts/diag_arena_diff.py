"""r5 diagnostic (how the conv3x3h race was found): three levels on three lanes, a fresh session per call; on a mismatch of the big level's map diff the model's whole workspace
against the reference run -- every eval activation has its own buffer in the arena, in execution order: the lowest differing offset names the first tensor that differs."""
import os, sys, numpy as np, torch, ctypes as C
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces.models.model import DetectionModel
from tinyfaces._hip import lib
from oracle.model import OracleDetectionModel, tame_init_
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
gen = torch.Generator().manual_seed(0)
sizes = [(350, 466), (700, 933), (1400, 1866)]
xs = [torch.randn(1, 3, h, w, generator=gen).cuda() for h, w in sizes]
R = int(os.environ.get("REPS", "60"))
H, W = sizes[2]
pb = lib().tf_detnet_param_region_bytes(1, 125, 0)
d2 = lambda v: (v - 1) // 2 + 1
H1, W1 = d2(H), d2(W); H2, W2 = d2(H1), d2(W1)
M1, M2 = H1 * W1, H2 * W2
al = lambda n: (n + 255) // 256 * 256
names, off = [], pb
def add(nm, nbytes):
    global off
    names.append((nm, off, off + nbytes)); off += al(nbytes)
add("partial", (max(((M1 + 63) // 64) * 2 * 64, 1100 * 3 * 1024) + 4096) * 4)
add("col", M1 * 192 * 2); add("cstem", M1 * 64 * 2); add("pool", M2 * 64 * 2)
h, w = H2, W2
blocks = [(64, False, 1)] + [(64, False, 1)] * 2 + [(128, True, 2)] + [(128, False, 1)] * 3 + [(256, True, 2)] + [(256, False, 1)] * 22
blocks[0] = (64, True, 1)
for i, (pl, ds, st) in enumerate(blocks):
    ho, wo = (d2(h), d2(w)) if st == 2 else (h, w)
    add(f"b{i}.c1", h * w * pl * 2); add(f"b{i}.c2", ho * wo * pl * 2)
    if ds: add(f"b{i}.d", ho * wo * pl * 4 * 2)
    add(f"b{i}.y", ho * wo * pl * 4 * 2)
    h, w = ho, wo
def where(o):
    for nm, a, b in names:
        if a <= o < b: return f"{nm}+{o - a}"
    return f"?{o}"
def run():
    with torch.no_grad(), m.constant_weights(reserve=(1, H, W)):
        outs = m.forward_levels(xs, lanes=3)
    torch.cuda.synchronize()
    return outs
ref = run(); ref_out = ref[2].clone(); ref_ws = m._ws.clone()
print("ws bytes", m._ws.numel(), "param region", pb, "modelled arena end", off, flush=True)
bad = 0
for it in range(R):
    outs = run()
    if not torch.equal(outs[2], ref_out):
        bad += 1
        diff = (m._ws != ref_ws)
        idx = torch.nonzero(diff).flatten()
        first, last, n = int(idx[0]), int(idx[-1]), int(idx.numel())
        # differing tensors in arena order
        hit = []
        for nm, a, b in names:
            c = int(diff[a:b].sum())
            if c: hit.append((nm, c))
            if len(hit) >= 6: break
        print("run", it, "differs: first byte", where(first), "last", where(last), "bytes", n, "first tensors:", hit, flush=True)
        if bad >= 4: break
print("mismatching runs", bad, flush=True)
