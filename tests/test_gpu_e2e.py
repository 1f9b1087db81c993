"""-m gpu: train -> checkpoint -> pyramid evaluation -> result files -> average precision, end to end on the GPU.

The reference's Makefile:11-21 is `python main.py ...` -> `python evaluate_model.py ... --checkpoint ...` -> the WIDER eval_tools.  Neither WIDER
images nor the MATLAB tool can reach this environment, so the loop is closed on a data set the detector can learn in seconds
(tinyfaces/datasets/synthetic.py: SyntheticFaces -- eight fixed 500x500 images with pasted faces): the two SCRIPTS are run as they are
(`main.py synthetic-faces ...`, `evaluate_model.py synthetic-faces --checkpoint ...`), the result files are read back by the native evaluator
(tinyfaces/wider_eval.py) against the boxes that were pasted, and the average precision must exceed 0.9.  One rank, and two gloo ranks sharing the
GPU (data-parallel training on disjoint shards + sharded evaluation)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tiny-faces-pytorch_amd")


def _run(cmd, cwd, env, timeout=400):
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    return r


def _ap(results_dir):
    from tinyfaces import wider_eval
    from tinyfaces.datasets.synthetic import SyntheticFaces
    from tinyfaces.datasets.templates import load_templates
    ds = SyntheticFaces(load_templates(), length=8, seed=0, train=False)
    gt = {"faces": ds.ground_truth()}
    keep = {"faces": {k: np.arange(v.shape[0]) for k, v in gt["faces"].items()}}
    raw = wider_eval.read_predictions(results_dir)
    assert sorted(raw["faces"]) == sorted(gt["faces"]), (sorted(raw.get("faces", {})), sorted(gt["faces"]))      # one result file per image
    ap, pr = wider_eval.evaluate_setting(wider_eval.norm_scores(raw), gt, keep)
    ndet = {k: int(v.shape[0]) for k, v in raw["faces"].items()}
    nface = sum(v.shape[0] for v in gt["faces"].values())
    return ap, ndet, nface


@pytest.mark.parametrize("ranks", [1, 2])
def test_train_checkpoint_evaluate_ap_on_pasted_faces(tmp_path, ranks):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    launch = [sys.executable]
    if ranks > 1:
        env.update(TINYFACES_DIST_BACKEND="gloo", TINYFACES_SHARE_GPU="1")
        launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                  "--master-port", str(29560 + ranks)]
    weights, results = str(tmp_path / "weights"), str(tmp_path / "results")
    # 300 steps of the fused engine at a constant learning rate: 3 epochs of 3200 samples in batches of 32 = the eight images x their four views
    # (the image itself and its x0.5 / x2 / x0.25 resamplings: what the evaluation pyramid will show), the same batch every step; two ranks: 16 each
    # (four layouts per rank, gradients averaged -- the same 300 global steps).  --ohem-thresh 0: with the reference's mining (defect D7) a positive
    # the classifier is sure of leaves the regression loss too, and the boxes of this from-scratch detector stay 10-20 pixels off (AP 0.3-0.8).
    bs = 32 // ranks
    r = _run(launch + [os.path.join(PKG, "main.py"), "synthetic-faces", "synthetic-faces", "--epochs", "3", "--synthetic-len", "3200", "--batch_size", str(bs),
                       "--lr", "2e-4", "--save-every", "3", "--save-path", weights, "--dtype", "bf16", "--ohem-thresh", "0"], str(tmp_path), env)
    ckpt = os.path.join(weights, "checkpoint_3.pth")
    assert os.path.exists(ckpt), r.stdout[-2000:]
    _run(launch + [os.path.join(PKG, "evaluate_model.py"), "synthetic-faces", "--checkpoint", ckpt, "--num-images", "8", "--results_dir", results,
                   "--prob_thresh", "0.1", "--mask-axis", "template"], str(tmp_path), env)
    ap, ndet, nface = _ap(results)
    report(f"learn_and_detect[ranks={ranks}]", ap=float(ap), faces=nface, detections=str(ndet))
    assert ap > 0.9, (ap, ndet, nface)


def test_bf16_fp16_detections_keep_the_oracles_index_set_on_separated_logits():
    """a detection-parity fixture for the bf16 / fp16 paths that CAN fail.  tests/planted.py builds a 960 x 1280 case whose logits are
    separated by construction (planted image, planted head on the tame-init trunk; table in tests/golden/planted_head.json).  The CPU oracle
    (oracle/pyramid.py = tinyfaces/evaluation.py:20-87) runs the 3-level pyramid (480 x 640, 960 x 1280, 1920 x 2560) once; the fp32, bf16 and fp16 HIP
    paths run `get_detections` at the same public threshold (prob 0.5, NMS 0.3).  Asserted:
      * the case is what it claims: with M = 2 x the LARGEST logit deviation measured here between any reduced-precision path and the oracle,
        `robustness` finds no survivor, candidate or overlap that an error of M could move (and M is far inside the planted gaps);
      * every path returns the oracle's candidate list (same count, same order, same boxes) and the IDENTICAL sorted keep set (`==`);
      * fp32 returns the oracle's keep ORDER as well (up to exact ties).
    A reduced-precision kernel whose logits drift by more than the planted margin (8 x the estimated bf16 rounding of a res3 channel) fails this."""
    import torch
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    from oracle import pyramid
    from oracle.model import OracleDetectionModel, tame_init_
    from oracle.nms import nms as onms
    from oracle.refstub import Compose as OCompose, Normalize as ONormalize, ToTensor as OToTensor
    from planted import load_plan, plant_head_, planted_image, robustness
    from tinyfaces import ops, transforms
    from tinyfaces.datasets.templates import load_templates
    from tinyfaces.evaluation import get_detections
    from tinyfaces.models.model import DetectionModel
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    templates = load_templates()
    plan = load_plan()
    om = plant_head_(tame_init_(OracleDetectionModel(num_templates=25), 0), plan).eval()
    sd = om.state_dict()
    u8 = planted_image()
    img = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
    otf = OCompose([OToTensor(), ONormalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    floor = 0.5                                                       # the oracle's list reaches half a logit below the threshold (far beyond M)
    _, allc, _ = pyramid.get_detections(om, img, templates, ops.RF, otf, prob_thresh=float(1 / (1 + np.exp(floor))), nms_thresh=0.3, scales=(-1, 0, 1),
                                        return_candidates=True)
    cand = allc[allc[:, 4] > 0]                                       # the decode is an order-preserving filter (utils.py:46)
    keep = onms(cand[:, :4], cand[:, 4], 0.3)
    assert cand.shape[0] >= 12 and len(keep) >= 8, (cand.shape, len(keep))
    assert len(keep) < cand.shape[0]                                  # the NMS has something to suppress
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    got = {}
    for name, dt, on_gpu in (("fp32", torch.float32, False), ("bf16", torch.bfloat16, False), ("fp16", torch.float16, False), ("bf16_gpu_pyramid", torch.bfloat16, True),
                             ("fp16_gpu_pyramid", torch.float16, True)):
        m = DetectionModel(num_templates=25)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().set_compute_dtype(dt).eval()
        got[name] = get_detections(m, img, templates, ops.RF, tf, prob_thresh=0.5, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", return_candidates=True,
                                   pyramid_on_gpu=on_gpu)
        if not on_gpu:                                                # the logit maps of the 960 x 1280 level against the oracle, same normalised input
            mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
            x = (img.unsqueeze(0) - mean) / std
            with torch.no_grad():
                got[name + "_err"] = float((m(x.cuda())[0, :25].float().cpu() - om(x)[0, :25]).abs().max())
        del m
    errs = {k[:-4]: v for k, v in got.items() if k.endswith("_err")}
    assert errs["fp32"] < 1e-3, errs                                  # north_star's fp32 bar
    for name in ("bf16", "fp16", "bf16_gpu_pyramid", "fp16_gpu_pyramid"):   # ... and at the candidates of all three levels
        c = got[name][1]
        if c.shape[0] == cand.shape[0]:
            errs[name + "@candidates"] = float(np.abs(c[:, 4] - cand[:, 4]).max())
    err = max(v for k, v in errs.items() if k != "fp32")
    M = 2 * err
    bad = robustness(cand, allc, keep, 0.0, M)
    report("separated_logits_detections", candidates=int(cand.shape[0]), survivors=int(len(keep)), logit_err=str({k: round(v, 5) for k, v in errs.items()}), margin=M,
           lowest_survivor_logit=float(cand[keep, 4].min()), highest_rejected_logit=float(allc[allc[:, 4] <= 0, 4].max()) if (allc[:, 4] <= 0).any() else -floor,
           violations=str(bad), kept=str({k: int(v[0].shape[0]) for k, v in got.items() if not k.endswith("_err")}))
    assert M < 0.16, errs                                             # the real error is inside what the fixture was planted for (K * EPS of make_planted_head.py)
    assert bad == (0, 0, 0, 0), (bad, M)
    for name in ("fp32", "bf16", "fp16", "bf16_gpu_pyramid", "fp16_gpu_pyramid"):
        d, c, k = got[name]
        assert c.shape[0] == cand.shape[0], (name, c.shape[0], cand.shape[0])
        assert np.abs(c[:, :4] - cand[:, :4]).max() < 1e-2, name       # the same candidates in the same order (the regression rows are zero: exact boxes)
        assert np.abs(c[:, 4] - cand[:, 4]).max() <= (1e-3 if name == "fp32" else err), name
        assert sorted(int(i) for i in k) == sorted(int(i) for i in keep), (name, k, keep)          # the IDENTICAL surviving index set
        assert np.array_equal(k, onms(c[:, :4], c[:, 4], 0.3)), name
        if name == "fp32":
            # fp32: the oracle's ORDER too, up to ties -- template 4 answers the flat top-right corner of the image with the same logit at all three
            # levels (0.18485704 / ..13 / ..37: a border response does not depend on the scale), and the two fp32 implementations differ by ~4e-7
            moved = np.nonzero(k != keep)[0]
            assert all(abs(cand[k[i], 4] - cand[keep[i], 4]) < 2e-6 for i in moved), (k, keep)
        assert d.shape[0] == len(keep)
