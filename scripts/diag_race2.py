"""r5 diagnostic, part 2: get_detections repeated on ONE image -- do the CANDIDATES (forward + decode) or only the NMS keep differ between runs?"""
import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces import ops, transforms
from tinyfaces.evaluation import get_detections
from tinyfaces.models.model import DetectionModel
from tinyfaces.datasets.templates import load_templates
from oracle.model import OracleDetectionModel, tame_init_
from oracle.targets import RF
templates = load_templates()
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().eval()
tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
gen = torch.Generator().manual_seed(0)
_ = torch.rand(3, 1875, 2500, generator=gen)
img2 = torch.rand(3, 700, 933, generator=gen)
R = int(os.environ.get("REPS", "30"))
for dt in (torch.bfloat16, torch.float16):
    m.set_compute_dtype(dt)
    x = torch.randn(1, 3, 937, 1250, generator=gen).cuda()
    with torch.no_grad():
        y = m(x).cpu()
    p = torch.sigmoid(y[0, :25]).flatten()
    thr = float(torch.quantile(p[:2000000], 0.9995))
    for gpu_pyr in (True, False):
        kw = dict(prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", pyramid_on_gpu=gpu_pyr, return_candidates=True)
        r0, c0, k0 = get_detections(m, img2, templates, RF, tf, **kw)
        nc = nk = 0; notes = []
        for _ in range(R):
            r, c, k = get_detections(m, img2, templates, RF, tf, **kw)
            if c.shape != c0.shape or not np.array_equal(c, c0):
                nc += 1
                if c.shape == c0.shape:
                    d = np.abs(c - c0); rows = np.nonzero(d.max(1) > 0)[0]
                    notes.append(("cand", len(rows), rows[:3].tolist(), d.max(0).round(5).tolist()))
                else:
                    notes.append(("cand-shape", c.shape, c0.shape))
            elif not np.array_equal(k, k0):
                nk += 1; notes.append(("keep", len(k), len(k0), np.setdiff1d(k, k0)[:4].tolist(), np.setdiff1d(k0, k)[:4].tolist()))
        print(dt, "pyramid_on_gpu", gpu_pyr, "candidates", c0.shape, "kept", k0.shape, "| runs with different candidates", nc, "| same candidates, different keep", nk, "of", R, notes[:4], flush=True)
        # NMS alone, on the fixed candidates
        b, s = torch.from_numpy(c0[:, :4].copy()).cuda(), torch.from_numpy(c0[:, 4].copy()).cuda()
        kk0 = ops.nms(b, s, 0.3).cpu().numpy(); bad = 0
        for _ in range(R):
            if not np.array_equal(ops.nms(b, s, 0.3).cpu().numpy(), kk0): bad += 1
        print("    nms alone on the fixed candidates: differing runs", bad, "of", R, flush=True)
