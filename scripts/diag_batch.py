import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces import _hip, transforms
from tinyfaces.evaluation import get_detections, get_detections_batch
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
img = torch.rand(3, 1875, 2500, generator=gen)
img2 = torch.rand(3, 700, 933, generator=gen)
for dt in (torch.float16, torch.bfloat16):
    m.set_compute_dtype(dt)
    x = torch.randn(1, 3, 937, 1250, generator=gen).cuda()
    with torch.no_grad():
        y = m(x).cpu()
    p = torch.sigmoid(y[0, :25]).flatten()
    thr = float(torch.quantile(p[:2000000], 0.9995))
    kw = dict(prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", pyramid_on_gpu=True)
    for rep in range(3):
        loop = [get_detections(m, im, templates, RF, tf, **kw) for im in (img, img2)]
        loop2 = [get_detections(m, im, templates, RF, tf, **kw) for im in (img, img2)]
        bat = get_detections_batch(m, [img, img2], templates, RF, tf, **kw)
        for i, (a, b, c) in enumerate(zip(loop, bat, loop2)):
            same_ll = a.shape == c.shape and np.array_equal(a, c)
            same_lb = a.shape == b.shape and np.array_equal(a, b)
            extra = ""
            if not same_lb and a.shape == b.shape:
                d = np.abs(a - b); extra = f" maxdiff {d.max():.3e} rows differing {int((d.max(1) > 0).sum())}"
            print(dt, "rep", rep, "img", i, "loop", a.shape, "batch", b.shape, "loop==loop", same_ll, "loop==batch", same_lb, extra, flush=True)
