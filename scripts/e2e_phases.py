"""Where the time of evaluation.get_detections(pyramid_on_gpu=True) goes for one 1280x960 host image (phases timed with synchronisations)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from bench import tame_init_
from tinyfaces import evaluation, ops, transforms
from tinyfaces.datasets.templates import load_templates
from tinyfaces.models.model import DetectionModel
dev = torch.device("cuda")
templates = load_templates()
torch.manual_seed(0)
model = tame_init_(DetectionModel(num_templates=25)).to(dev).set_compute_dtype("bf16").eval()
rs = np.random.RandomState(11)
base = rs.randint(0, 256, (60, 80, 3)).astype(np.uint8)
u8 = np.kron(base, np.ones((16, 16, 1), np.uint8)) ^ rs.randint(0, 32, (960, 1280, 3)).astype(np.uint8)
img = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
tfm = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
def T():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad(), model.constant_weights(reserve=(1, 1920, 2560)):
    for rep in range(4):
        t0 = T(); px = transforms.to_uint8_hwc(img)
        t1 = T(); d = torch.from_numpy(px).to(dev)
        t2 = T(); levels = evaluation._pyramid_levels(img, (-1, 0, 1), tfm, True, dev)
        t3 = T(); t_d = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(dev)
        nt = 25
        dets = torch.empty(max(evaluation._level_capacity(levels, nt), 1), 5, dtype=torch.float64, device=dev); count = torch.zeros(1, dtype=torch.int32, device=dev)
        t4 = T(); evaluation._decode_levels(model, levels, templates, t_d, ops.RF, 0.9, "w", dets, count, dev)
        t5 = T(); n = int(count.item()); cand = dets[:n]; keep = ops.nms(cand[:, :4].contiguous(), cand[:, 4].contiguous(), 0.3)
        t6 = T(); r = cand[keep].cpu().numpy()
        t7 = T(); full = evaluation.get_detections(model, img, templates, ops.RF, tfm, prob_thresh=0.9, nms_thresh=0.3, scales=(-1, 0, 1), device=dev, pyramid_on_gpu=True)
        t8 = T()
        print(f"[{rep}] quantise {1e3*(t1-t0):.1f}  upload {1e3*(t2-t1):.1f}  _pyramid_levels (quantise+upload+3 prepares) {1e3*(t3-t2):.1f}  templates+alloc {1e3*(t4-t3):.1f}  "
              f"forwards+decode {1e3*(t5-t4):.1f}  nms(n={n}) {1e3*(t6-t5):.1f}  rows back {1e3*(t7-t6):.1f}  | get_detections {1e3*(t8-t7):.1f} ms", flush=True)
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
