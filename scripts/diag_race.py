"""r5 diagnostic: where does the evaluation path stop being bit-reproducible?  (a) eval forward of ONE level repeated; (b) the three levels side by
side on the model's lanes (forward_levels) against the sequential loop; (c) decode + NMS on FIXED maps."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces import ops
from tinyfaces.models.model import DetectionModel
from tinyfaces.datasets.templates import load_templates
from oracle.model import OracleDetectionModel, tame_init_
templates = load_templates()
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().eval()
gen = torch.Generator().manual_seed(0)
sizes = [(350, 466), (700, 933), (1400, 1866)]
xs = [torch.randn(1, 3, h, w, generator=gen).cuda() for h, w in sizes]
R = int(os.environ.get("REPS", "12"))
for dt in (torch.bfloat16, torch.float16):
    m.set_compute_dtype(dt)
    with torch.no_grad(), m.constant_weights(reserve=(1, 1400, 1866)):
        ref = [m(x).clone() for x in xs]
        torch.cuda.synchronize()
        for i, x in enumerate(xs):
            bad = 0; worst = 0.0
            for _ in range(R):
                y = m(x)
                if not torch.equal(y, ref[i]):
                    bad += 1; worst = max(worst, float((y - ref[i]).abs().max()))
            print(dt, "single level", sizes[i], "mismatching repeats", bad, "of", R, "max diff", worst, flush=True)
        for lanes in (1, 3):
            bad = [0, 0, 0]; worst = [0.0] * 3; where = {}
            for _ in range(R):
                outs = m.forward_levels(xs, lanes=lanes)
                torch.cuda.synchronize()
                for i, y in enumerate(outs):
                    if not torch.equal(y, ref[i]):
                        bad[i] += 1; d = (y - ref[i]).abs(); worst[i] = max(worst[i], float(d.max()))
                        idx = torch.nonzero(d.amax(dim=(0, 1)) > 0)
                        where.setdefault(i, []).append((int(idx.shape[0]), idx[:3].tolist()))
            print(dt, f"forward_levels lanes={lanes}: mismatching repeats per level", bad, "max diff", worst, {k: v[:2] for k, v in where.items()}, flush=True)
