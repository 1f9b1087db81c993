#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN SOURCE (/root/reference)
in this container.  Container-only: the GPU box has no /root/reference; the committed
fixtures (data only: inputs + expected outputs) are what travels.

  python oracle/tools/make_golden.py            # (re)write fixtures
  python oracle/tools/make_golden.py --check    # regenerate in memory, compare with committed

torchvision / pyclust / pyclustering are absent here, so oracle/refstub.py installs stubs
first (the resnet101 trunk, nms and transforms behind those stubs are this repo's
restatements of third-party code; everything else executed below is reference source).
"""
import argparse
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refstub  # noqa: E402

refstub.install()
from tinyfaces.datasets.dense_overlap import compute_dense_overlap  # noqa: E402  (reference)
from tinyfaces.datasets.processor import DataProcessor  # noqa: E402
from tinyfaces.models.loss import DetectionCriterion  # noqa: E402
from tinyfaces.models.model import DetectionModel  # noqa: E402
from tinyfaces.models.utils import get_bboxes  # noqa: E402
from tinyfaces import evaluation as ref_eval  # noqa: E402
from tinyfaces import trainer as ref_trainer  # noqa: E402

from oracle.model import tame_init_  # noqa: E402  (init recipe only)
from oracle.nms import nms as oracle_nms  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
RF = {"size": [859, 859], "stride": [8, 8], "offset": [-1, -1]}


def load_templates():
    import json
    t = json.load(open(os.path.join(refstub.REFERENCE_ROOT, "tinyfaces/datasets/templates.json")))
    return np.round(np.array(t), decimals=8)      # datasets/__init__.py:38


def random_boxes(rng, g, size=500):
    """(x1,y1,x2,y2) f64 boxes; includes exact-template-size, 1-px, out-of-map, degenerate."""
    w = np.exp(rng.uniform(np.log(8), np.log(200), g))
    h = w * rng.uniform(1.0, 1.5, g)
    x1 = rng.uniform(-20, size - 10, g)
    y1 = rng.uniform(-20, size - 10, g)
    b = np.stack([x1, y1, x1 + w, y1 + h], 1)
    return np.round(b, 3)


def gen_targets(out):
    templates = load_templates()
    out["templates"] = templates
    cases = []
    rng = np.random.RandomState(1234)
    specials = np.array([
        [100.0, 100.0, 100.0 + 2 * 20.0, 100.0 + 2 * 22.5],          # exactly template 9's size
        [250.0, 250.0, 251.0, 251.0],                                # ~1 px
        [600.0, 600.0, 700.0, 720.0],                                # outside the map
        [50.0, 60.0, 50.0, 90.0],                                    # degenerate (x2<=x1): dropped
        [7.0, 7.0, 7.0 + 2 * 9.6613, 7.0 + 2 * 10.6161],             # template 24 centred on a cell
    ])
    paste_boxes = [[0, 0, 500, 500], [91, 193, 491, 493], [0, 120, 500, 380]]
    for ci, g in enumerate([0, 1, 3, 16, 5]):
        boxes = random_boxes(rng, g) if ci < 4 else specials
        paste = paste_boxes[ci % 3]
        seed = 100 + ci
        proc = DataProcessor((500, 500), (63, 63), 0.7, 0.3, templates, rf=RF)
        pad_mask = proc.get_padding(paste)
        flip = bool(ci % 2)
        if flip:
            pad_mask = np.fliplr(pad_mask)                            # wider_face.py:165
        np.random.seed(seed)
        cm, rm, iou_pert = proc.get_heatmaps(boxes.copy(), pad_mask)
        # the only np.random draw inside get_heatmaps is rand(*iou.shape) (processor.py:195)
        valid = ~np.logical_or(boxes[:, 2] <= boxes[:, 0], boxes[:, 3] <= boxes[:, 1])
        ngv = int(valid.sum())
        np.random.seed(seed)
        noise = np.random.rand(63, 63, 25, ngv) if ngv > 0 else np.zeros((63, 63, 25, 0))
        bv = boxes[valid]
        iou = compute_dense_overlap(-1, -1, 8, 8, 63, 63, templates[:, 0], templates[:, 1],
                                    templates[:, 2], templates[:, 3], bv[:, 0], bv[:, 1],
                                    bv[:, 2], bv[:, 3], 1, 1) if ngv else np.zeros((63, 63, 25, 0))
        tag = f"c{ci}"
        out[f"{tag}_boxes"] = boxes
        out[f"{tag}_paste"] = np.array(paste, dtype=np.int64)
        out[f"{tag}_flip"] = np.array(int(flip))
        out[f"{tag}_pad"] = np.packbits(pad_mask)
        # noise is NOT stored: tests regenerate it with np.random.seed(seed); np.random.rand(63,63,25,ngv)
        out[f"{tag}_seed"] = np.array(seed)
        out[f"{tag}_class"] = cm.astype(np.int8)
        full = ng_full = ci in (1, 4)
        # fixtures stay small: full f64 maps for two cases, a strided sample + checksum for the rest
        out[f"{tag}_full"] = np.array(int(full))
        out[f"{tag}_iou"] = iou if full else iou[::4, ::4]
        out[f"{tag}_reg"] = rm if full else rm[::4, ::4]
        out[f"{tag}_iou_sum"] = np.array(iou.sum())
        out[f"{tag}_reg_sum"] = np.array(rm.sum())
        out[f"{tag}_ioupert_max"] = iou_pert.max(axis=(0, 1, 2)) if ngv else np.zeros((0,))
        cases.append(tag)
    out["cases"] = np.array(cases)


def gen_decode(out):
    templates = load_templates()
    cases = []
    rng = np.random.RandomState(77)
    for ci, (h, w, scale) in enumerate([(9, 25, 0.5), (12, 40, 1), (20, 80, 2), (7, 31, 1)]):
        score_cls = (rng.randn(1, h, w, 25) * 2.0).astype(np.float32)
        score_reg = (rng.randn(1, h, w, 100) * 0.5).astype(np.float32)
        prob = 1.0 / (1.0 + np.exp(-score_cls.astype(np.float64)))
        prob = prob.astype(np.float32)
        thr = 0.8
        b, s = get_bboxes(score_cls, score_reg, prob.copy(), templates, thr, RF, scale)
        tag = f"d{ci}"
        out[f"{tag}_score_cls"], out[f"{tag}_score_reg"], out[f"{tag}_prob"] = score_cls, score_reg, prob
        out[f"{tag}_scale"], out[f"{tag}_thr"] = np.array(float(scale)), np.array(thr)
        out[f"{tag}_boxes"], out[f"{tag}_scores"] = b, s
        # refine=False (utils.py:65-66): what the reference ACTUALLY returns -- `bboxes[0]` of a (4, N) array, the x1 row
        b0, s0 = get_bboxes(score_cls, score_reg, prob.copy(), templates, thr, RF, scale, refine=False)
        out[f"{tag}_norefine"], out[f"{tag}_norefine_scores"] = np.asarray(b0, dtype=np.float64), s0
        cases.append(tag)
    out["cases"] = np.array(cases)
    # W < 25 -> the reference raises IndexError (defect D1, utils.py:44)
    raised = False
    try:
        z = np.zeros((1, 5, 20, 25), np.float32)
        get_bboxes(z, np.zeros((1, 5, 20, 100), np.float32), z.copy(), templates, 0.5, RF, 1)
    except IndexError:
        raised = True
    out["w_lt_25_raises_indexerror"] = np.array(int(raised))


def gen_criterion(out):
    cases = []
    for ci, (bsz, h, w, seed, npos, nneg) in enumerate([(2, 9, 11, 5, 40, 300), (3, 16, 16, 6, 400, 4000),
                                                        (1, 8, 8, 7, 0, 20), (1, 40, 40, 8, 300, 30000)]):
        g = torch.Generator().manual_seed(seed)
        output = torch.randn(bsz, 125, h, w, generator=g) * 1.5
        cm = torch.zeros(bsz, 25 * h * w)
        for b in range(bsz):
            perm = torch.randperm(25 * h * w, generator=g)
            n_p, n_n = min(npos, perm.numel() // 3), min(nneg, perm.numel() // 2)
            cm[b, perm[:n_p]] = 1
            cm[b, perm[n_p:n_p + n_n]] = -1
        cm = cm.view(bsz, 25, h, w)
        rmap = torch.randn(bsz, 100, h, w, generator=g) * 0.7
        crit = DetectionCriterion(25)
        o = output.clone().requires_grad_(True)
        cm_in = cm.clone()
        np.random.seed(seed)
        loss = crit(o, cm_in, rmap)
        loss.backward()
        tag = f"k{ci}"
        out[f"{tag}_output"], out[f"{tag}_class_map"], out[f"{tag}_reg_map"] = output.numpy(), cm.numpy().astype(np.int8), rmap.numpy()
        out[f"{tag}_seed"] = np.array(seed)
        out[f"{tag}_loss"] = np.array([float(loss), float(crit.masked_class_loss.sum()), float(crit.masked_reg_loss.sum())])
        out[f"{tag}_class_after_ohem_inplace"] = cm_in.numpy().astype(np.int8)     # loss.py:62 mutates the input
        out[f"{tag}_grad"] = o.grad.numpy()
        out[f"{tag}_avg"] = np.array([crit.class_average.average, crit.reg_average.average])
        cases.append(tag)
    out["cases"] = np.array(cases)


def gen_nms(out):
    """torchvision.ops.nms is third-party and absent: these vectors come from the RESTATED
    kernel (oracle/nms.py) -- flagged 3p-restated / parity unpinned."""
    rng = np.random.RandomState(9)
    cases = []
    for ci, n in enumerate([0, 1, 257, 1251, 4096]):
        cx, cy = rng.uniform(0, 800, n), rng.uniform(0, 600, n)
        w = np.exp(rng.uniform(np.log(8), np.log(120), n))
        h = w * rng.uniform(1.0, 1.4, n)
        boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        scores = rng.randn(n).astype(np.float32).astype(np.float64)
        if n >= 257:   # adversarial: ties, identical boxes, zero-area boxes
            scores[10:20] = scores[10]
            boxes[30:35] = boxes[30]
            boxes[40, 2] = boxes[40, 0]
            boxes[41] = [5.0, 5.0, 5.0, 5.0]
            boxes[42] = [5.0, 5.0, 5.0, 5.0]
        keep = oracle_nms(boxes, scores, 0.3)
        tag = f"n{ci}"
        out[f"{tag}_boxes"], out[f"{tag}_scores"], out[f"{tag}_keep"] = boxes, scores, keep
        cases.append(tag)
    out["cases"] = np.array(cases)


def ref_model(seed=0):
    m = DetectionModel(num_objects=1, num_templates=25)      # reference class over the stub trunk
    tame_init_(m, seed)
    return m


def gen_model(out):
    cases = []
    m = ref_model(0)
    keys = list(m.state_dict().keys())
    out["state_dict_keys"] = np.array(keys)
    out["state_dict_numel"] = np.array([v.numel() for v in m.state_dict().values()], dtype=np.int64)
    out["bilinear_diag"] = m.score4_upsample.weight[3, 3].detach().numpy()
    out["bilinear_offdiag_abs_sum"] = np.array(float(m.score4_upsample.weight.detach().abs().sum()
                                                     - 125 * m.score4_upsample.weight[0, 0].abs().sum()))
    lp = m.learnable_parameters(0.5)
    out["group_sizes"] = np.array([len(list(g["params"])) for g in lp])
    out["group_lrs"] = np.array([g["lr"] for g in m.learnable_parameters(0.5)], dtype=np.float64)
    for ci, (b, h, w, train) in enumerate([(1, 64, 64, False), (1, 97, 131, False), (2, 96, 128, True),
                                           (1, 120, 200, True)]):
        g = torch.Generator().manual_seed(50 + ci)
        x = torch.randn(b, 3, h, w, generator=g)
        m = ref_model(0)
        m.train(train)
        xin = x.clone().requires_grad_(True)
        y = m(xin)
        tag = f"m{ci}"
        out[f"{tag}_x"], out[f"{tag}_train"], out[f"{tag}_y"] = x.numpy(), np.array(int(train)), y.detach().numpy()
        gy = torch.randn(y.shape, generator=g) * (torch.rand(y.shape, generator=g) < 0.02)
        y.backward(gy)
        out[f"{tag}_gy"] = gy.numpy()
        out[f"{tag}_gx"] = xin.grad.numpy()
        for name in ("score_res3.weight", "score_res4.bias", "model.layer3.22.conv3.weight",
                     "model.layer2.0.downsample.0.weight", "model.layer1.0.bn1.weight", "model.conv1.weight"):
            p = dict(m.named_parameters())[name]
            if p.numel() > 20000 and ci in (1, 3):
                continue                                  # keep the fixture small
            out[f"{tag}_grad::{name}"] = p.grad.numpy()
        if train:
            out[f"{tag}_rm::model.bn1.running_mean"] = m.model.bn1.running_mean.numpy()
            out[f"{tag}_rv::model.layer3.5.bn2.running_var"] = m.model.layer3[5].bn2.running_var.numpy()
        cases.append(tag)
    out["cases"] = np.array(cases)


def gen_detections(out):
    templates = load_templates()
    m = ref_model(0)
    g = torch.Generator().manual_seed(321)
    img = torch.rand(3, 300, 400, generator=g)
    tf = refstub.Compose([refstub.ToTensor(), refstub.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    with torch.no_grad():   # random weights give no WIDER-like sparsity: calibrate thr as a quantile
        m.eval()
        y0 = m(tf(refstub.to_pil_image(img)).unsqueeze(0))
        thr = float(np.round(np.quantile(torch.sigmoid(y0[:, :25]).numpy(), 0.97), 4))
    with torch.no_grad():
        dets = ref_eval.get_detections(m, img, templates, RF, tf, prob_thresh=thr, nms_thresh=0.3,
                                       scales=(-1, 0, 1), device=torch.device("cpu"))
    out["img"], out["thr"], out["dets_ref_K4"] = img.numpy(), np.array(thr), dets
    out["scales"] = np.array([-1, 0, 1])


def gen_trainer(out):
    """2 steps at bs=2 through the reference trainer.train (trainer.py:68-90)."""
    m = ref_model(0)
    crit = DetectionCriterion(25)
    opt = torch.optim.SGD(m.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    g = torch.Generator().manual_seed(11)
    batches = []
    for _ in range(2):
        img = torch.randn(2, 3, 128, 128, generator=g)
        cm = (torch.randint(0, 40, (2, 25, 16, 16), generator=g) == 0).double() - \
             (torch.randint(0, 3, (2, 25, 16, 16), generator=g) == 0).double()
        cm = cm.clamp(-1, 1)
        rm = torch.randn(2, 100, 16, 16, generator=g).double() * 0.5
        batches.append((img, cm, rm))
    np.random.seed(3)
    buf = io.StringIO()
    with redirect_stdout(buf):
        ref_trainer.train(m, crit, opt, batches, 0, torch.device("cpu"))
    out["log"] = np.array(buf.getvalue())
    for i, (img, cm, rm) in enumerate(batches):
        out[f"b{i}_img"], out[f"b{i}_cm"], out[f"b{i}_rm"] = img.numpy(), cm.numpy().astype(np.int8), rm.numpy()
    sd = m.state_dict()
    out["avg"] = np.array([crit.class_average.average, crit.reg_average.average])
    for name in ("score_res3.weight", "score_res4.weight", "model.conv1.weight", "model.layer3.22.conv3.weight",
                 "model.layer2.1.bn2.weight", "model.bn1.running_mean", "model.layer3.0.bn3.running_var"):
        out[f"after::{name}"] = sd[name].numpy()
    out["param_checksum"] = np.array(float(sum(v.double().sum() for k, v in sd.items() if v.is_floating_point())))


def synth_image(seed, H, W):
    """Smooth + noisy uint8 RGB test image (compresses well, still exercises every resample tap); regenerated by the tests
    from (seed, H, W) so the fixture stores only the reference's OUTPUTS."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 110 * np.sin(xx / (9.0 + c) + c) * np.cos(yy / (6.0 + 2 * c)) for c in range(3)], 2)
    img = img + rng.randint(-2, 3, (H, W, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def gen_augment(out):
    """WIDERFace.process_inputs (tinyfaces/datasets/wider_face.py:133-192) + DataProcessor.crop_image
    (tinyfaces/datasets/processor.py:41-112) of the reference, PIL behind the torchvision resize stub.
    Also a direct PIL resize vector for the eval pyramid (tinyfaces/evaluation.py:46)."""
    from PIL import Image
    from tinyfaces.datasets.wider_face import WIDERFace
    templates = load_templates()
    proc = DataProcessor((500, 500), (63, 63), 0.7, 0.3, templates, rf=RF)
    ds = object.__new__(WIDERFace)                       # no dataset on disk: only the fields process_inputs reads
    ds.processor, ds.input_size, ds.debug, ds.templates = proc, (500, 500), False, templates
    cases = [(3, 700, 900), (4, 300, 400), (7, 1024, 768), (9, 520, 480), (12, 260, 1200), (21, 499, 501)]
    out["cases"] = np.array(cases, dtype=np.int64)
    for n, (seed, H, W) in enumerate(cases):
        img = synth_image(seed, H, W)
        rng = np.random.RandomState(100 + seed)
        boxes = random_boxes(rng, 12, size=max(H, W))
        out[f"a{n}_boxes_in"] = boxes
        np.random.seed(1000 + seed)
        res_img, cm, rm, bb = ds.process_inputs(Image.fromarray(img, "RGB"), boxes.copy())
        out[f"a{n}_img"] = res_img
        out[f"a{n}_boxes"] = np.asarray(bb, dtype=np.float64).reshape(-1, 4)
        out[f"a{n}_pos"] = np.int64((cm == 1).sum())      # a checksum of the maps that follow (same random stream)
    # WIDERFace.load (wider_face.py:65-121) on a hand-made annotation file: empty image (placeholder line), zero-size boxes,
    # negative numbers, trailing attributes
    import tempfile
    ann = ("0--Parade/a.jpg\n3\n10 20 30 40 0 0 0 0 0 0\n-5 7 0 12 1 0 0 1 0 0\n100.5 50 8 9 2 1 1 0 2 1\n"
           "1--Handshaking/b.jpg\n0\n0 0 0 0 0 0 0 0 0 0 \n"
           "2--Demo/c.jpg\n2\n1 1 1 1 0 0 0 0 0 0\n300 200 25 0 0 0 0 0 0 0\n")
    out["ann_text"] = np.array(ann)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(ann)
    ds2 = object.__new__(WIDERFace)
    ds2.split = "train"
    ds2.load(f.name)
    os.unlink(f.name)
    out["ann_paths"] = np.array([d["img_path"] for d in ds2.data])
    for i, d in enumerate(ds2.data):
        out[f"ann{i}_bboxes"] = np.asarray(d["bboxes"], dtype=np.float64).reshape(-1, 4)
        out[f"ann{i}_attrs"] = np.stack([d[k] for k in ("blur", "expression", "illumination", "invalid", "occlusion", "pose")]).astype(np.float64)
    # evaluation.py:46: transforms.functional.resize(image, int(min_side * scale)) for scales 2^-1 .. 2^1 of a 180x240 image
    img = synth_image(5, 180, 240)
    for k, s in enumerate((0.5, 2)):
        r = refstub.resize(Image.fromarray(img, "RGB"), int(180 * s))
        out[f"r{k}_img"] = np.array(r)


def gen_clustering(out):
    """tinyfaces/clustering: centralize_bbox + compute_distances (cluster.py:13-37) and the reference's own k-medoids
    (k_medoids.py, option 'local') under a fixed np.random seed."""
    from tinyfaces.clustering.cluster import centralize_bbox, compute_distances
    from tinyfaces.clustering.k_medoids import kMedoids
    rng = np.random.RandomState(5)
    boxes = random_boxes(rng, 90, size=800)
    out["boxes"] = boxes
    with redirect_stdout(io.StringIO()):
        shapes = centralize_bbox(boxes)
        dist = compute_distances(shapes)
    out["shapes"], out["dist"] = shapes, dist
    for k in (3, 7):
        np.random.seed(40 + k)
        med, member = kMedoids(dist, k)
        out[f"k{k}_medoids"], out[f"k{k}_member"] = np.asarray(med, dtype=np.int64), np.asarray(member, dtype=np.int64)


def gen_cli(out):
    """The command lines of the reference's entry scripts (main.py:18-36, evaluate_model.py:17-31): every option with its default
    as argparse resolves them for a minimal positional-only invocation."""
    import importlib.util
    import json

    def defaults(path, argv):
        spec = importlib.util.spec_from_file_location("ref_cli_" + os.path.basename(path)[:-3], path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        old = sys.argv
        sys.argv = [path] + argv
        try:
            return vars(mod.arguments())
        finally:
            sys.argv = old
    out["main"] = np.array(json.dumps(defaults(os.path.join(refstub.REFERENCE_ROOT, "main.py"), ["TRAIN", "VAL"]), sort_keys=True))
    out["evaluate_model"] = np.array(json.dumps(defaults(os.path.join(refstub.REFERENCE_ROOT, "evaluate_model.py"), ["DATA"]), sort_keys=True))


GENERATORS = {"cli": gen_cli, "clustering": gen_clustering, "targets": gen_targets, "decode": gen_decode, "criterion": gen_criterion, "nms": gen_nms,
              "model": gen_model, "detections": gen_detections, "trainer": gen_trainer, "augment": gen_augment}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    bad = 0
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only.split(","):
            continue
        out = {}
        fn(out)
        path = os.path.join(GOLD, f"{name}.npz")
        if args.check:
            old = np.load(path, allow_pickle=False)
            for k, v in out.items():
                a, b = np.asarray(v), old[k]
                same = (a.shape == b.shape) and (np.array_equal(a, b) if a.dtype.kind in "iubUS" else
                                                 np.allclose(a, b, rtol=1e-6, atol=1e-7, equal_nan=True))
                if not same:
                    bad += 1
                    print(f"MISMATCH {name}:{k}")
            print(f"checked {name}: {len(out)} arrays")
        else:
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
