"""r5 diagnostic: is a training step bit-reproducible where it should be?  Reproducible BN statistics (tf_set_stat_rows(0): no atomics in the statistic sums);
the weight gradients of the layer-3 identity bottlenecks are full-reduction kernels with plain stores, so with a deterministic data-gradient chain their slice
of the flat gradient must repeat itself bit for bit -- beside the real two-stream contention of the backward pass."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces import _hip
from tinyfaces.models.model import DetectionModel
from oracle.model import OracleDetectionModel, tame_init_
lib = _hip.lib()
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().set_compute_dtype(torch.bfloat16).train()
m.model.bn1.momentum = 0.0
flat = m.flatten_parameters()
g = torch.Generator().manual_seed(1)
x = torch.randn(12, 3, 500, 500, generator=g).cuda()
lib.tf_set_stat_rows(0)
m._sync_tables(x.device)
R = int(os.environ.get("REPS", "12"))
seg = m._segments
def rng(prefix):
    ks = [k for k in seg if k.startswith(prefix)]
    return min(seg[k][0] for k in ks), max(seg[k][0] + seg[k][1] for k in ks)
ref = None
names = [f"model.layer3.{i}." for i in range(1, 23)]
bad_out = bad_grad = 0; where = {}
for it in range(R + 1):
    out = m._run_forward(x, training=True)
    gy = torch.full_like(out, 1e-3)
    gf = m._run_backward(x, gy, persistent=True)
    torch.cuda.synchronize()
    cur = (out.clone(), gf.clone())
    if ref is None:
        ref = cur; continue
    if not torch.equal(cur[0], ref[0]): bad_out += 1
    diff = cur[1] != ref[1]
    if bool(diff.any()):
        bad_grad += 1
        for k, (o, n) in seg.items():
            c = int(diff[o:o + n].sum())
            if c: where[k] = where.get(k, 0) + 1
print("runs", R, "| forward maps differ:", bad_out, "| flat gradient differs:", bad_grad, flush=True)
l3 = {k: v for k, v in where.items() if any(k.startswith(p) for p in names)}
print("tensors of the layer-3 identity bottlenecks that ever differed:", len(l3), sorted(l3.items())[:12], flush=True)
print("other tensors that ever differed:", len(where) - len(l3), sorted((k, v) for k, v in where.items() if k not in l3)[:16], flush=True)
