"""r5 diagnostic, part 5: get_detections repeated -- with the level lanes off, and inside ONE outer constant_weights session."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces import transforms
from tinyfaces.evaluation import get_detections
from tinyfaces.models.model import DetectionModel
from tinyfaces.datasets.templates import load_templates
from oracle.model import OracleDetectionModel, tame_init_
from oracle.targets import RF
import contextlib
templates = load_templates()
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
gen = torch.Generator().manual_seed(0)
_ = torch.rand(3, 1875, 2500, generator=gen)
img2 = torch.rand(3, 700, 933, generator=gen)
R = int(os.environ.get("REPS", "40"))
x = torch.randn(1, 3, 937, 1250, generator=gen).cuda()
with torch.no_grad():
    y = m(x).cpu()
p = torch.sigmoid(y[0, :25]).flatten()
thr = float(torch.quantile(p[:2000000], 0.9995))
kw = dict(prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", pyramid_on_gpu=True, return_candidates=True)
for tag, lanes, outer in (("lanes=3, session per call", "3", False), ("lanes=1", "1", False), ("lanes=3 inside one outer session", "3", True), ("lanes=2", "2", False)):
    os.environ["TINYFACES_EVAL_LANES"] = lanes
    ctx = m.constant_weights(reserve=(1, 1400, 1866)) if outer else contextlib.nullcontext()
    with ctx:
        r0, c0, k0 = get_detections(m, img2, templates, RF, tf, **kw)
        bad = 0
        for _ in range(R):
            r, c, k = get_detections(m, img2, templates, RF, tf, **kw)
            if c.shape != c0.shape or not np.array_equal(c, c0): bad += 1
    print(tag, "-> runs with different candidates", bad, "of", R, flush=True)
