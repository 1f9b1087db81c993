"""A detection case with SEPARATED logits, for the reduced-precision detection-parity test (tests/test_gpu_e2e.py).

With random weights every survivor of the pyramid sits within twice the bf16 error of the threshold, and a trained detector answers a face with several
templates and positions whose scores are closer than that error: on neither can a bf16 path be held to the oracle's surviving INDEX set.  Here the logits
are separated by construction:

* the image is a flat grey field with a dozen checker patches of graded contrast (`planted_image`), so a feature channel of the (random, tame-init)
  trunk answers with a few isolated peaks of clearly different height;
* the head is planted (`plant_head_`): score_res4 is zero, the regression rows are zero (every box is its template at its position), and the
  classification row of each of the eight templates the decode ever admits (4..11, tinyfaces/models/utils.py:17-41) reads ONE res3 channel,
  logit = gain * (res3[channel] - tau), with the gain set so that every row carries about the same bf16 error and tau inside a wide gap of that
  channel's sorted values over the three pyramid levels.  All other rows sit at -30.  The table (template, channel, tau, gain) is tests/golden/planted_head.json, found by oracle/tools/make_planted_head.py.

`robustness` is the check that the case is what it claims: against a score error of M, no survivor can fall below the threshold, no suppressed
candidate can rise above it or overtake the survivor that suppresses it, and no suppression hangs on an IoU near the NMS threshold.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
VALID_TEMPLATES = tuple(range(4, 12))


def planted_image(seed=0, H=960, W=1280, k=12, decay=0.93):
    rs = np.random.RandomState(seed)
    img = np.full((H, W, 3), 115, np.uint8)
    cells = [(r, c) for r in range(4) for c in range(6)]
    rs.shuffle(cells)
    for i, (r, c) in enumerate(cells[:k]):
        s = (32, 48, 64, 96)[i % 4]
        y, x = 40 + r * 230 + rs.randint(0, 60), 40 + c * 205 + rs.randint(0, 60)
        patch = (rs.rand(s // 8, s // 8, 3) - 0.5) * 2 * decay ** i
        img[y:y + s, x:x + s] = np.clip(115 + 115 * np.kron(patch, np.ones((8, 8, 1))), 0, 255).astype(np.uint8)
    return img


def load_plan():
    with open(os.path.join(HERE, "golden", "planted_head.json")) as f:
        return json.load(f)


def plant_head_(model, plan, num_templates=25):
    """plan = {"rows": [[template, channel, tau, gain], ...]}; `model` has score_res3 / score_res4 (the oracle's or the product's)."""
    import torch
    with torch.no_grad():
        model.score_res4.weight.zero_()
        model.score_res4.bias.zero_()
        model.score_res3.weight.zero_()
        model.score_res3.bias.zero_()
        model.score_res3.bias[:num_templates] = -30.0
        for t, c, tau, gain in plan["rows"]:
            model.score_res3.weight[t, c, 0, 0] = gain
            model.score_res3.bias[t] = -gain * tau
    return model


def iou_matrix(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    area = lambda r: (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
    return inter / (area(a)[:, None] + area(b)[None, :] - inter)


def robustness(cand, allc, keep, thr, M, nms_thresh=0.3, band=0.03):
    """cand (N, 5): the candidates above `thr` (logit) in list order, keep: NMS survivors (indices into cand); allc (K, 5): every position the decode
    could admit with its logit (a superset reaching well below thr).  Returns the four violation counts (all zero = the keep set cannot move under a
    score error of M):
      near      survivors within M of the threshold, or sub-threshold positions within M of it (the candidate LIST itself could change),
      contested survivors led by less than M over a candidate they overlap by more than nms_thresh - band,
      uncovered suppressed candidates that no survivor suppresses for certain (IoU > nms_thresh + band with a lead > M),
      borderline (survivor, candidate) pairs whose IoU is within `band` of the NMS threshold."""
    surv = cand[keep]
    near = int((surv[:, 4] - thr <= M).sum()) + int(((allc[:, 4] <= thr) & (allc[:, 4] > thr - M)).sum()) + int(((cand[:, 4] - thr) <= M).sum())
    iou = iou_matrix(surv, cand)
    lead = surv[:, 4][:, None] - cand[:, 4][None, :]
    me = np.zeros_like(iou, bool)
    me[np.arange(len(keep)), keep] = True
    contested = int(((iou > nms_thresh - band) & ~me & (lead <= M)).any(axis=1).sum())
    covered = ((iou > nms_thresh + band) & (lead > M)).any(axis=0) | me.any(axis=0)
    uncovered = int((~covered).sum())
    borderline = int(((np.abs(iou - nms_thresh) < band) & ~me).sum())
    return near, contested, uncovered, borderline
