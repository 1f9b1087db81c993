"""Restatement of `torchvision.ops.nms` CPU kernel semantics (torchvision 0.18,
csrc/ops/cpu/nms_kernel.cpp -- third-party, NOT in /root/reference; PARITY UNPINNED).

Call site in the reference: tinyfaces/evaluation.py:84 with float64 boxes/scores.

Semantics restated:
  areas = (x2-x1)*(y2-y1)                      (no +1)
  order = argsort(scores, descending, STABLE)
  for i in order: if not suppressed[i]: keep i; for every later j in order:
        inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1))
        ovr   = inter / (area_i + area_j - inter);  suppressed[j] |= ovr > thr   (strict)
  returns kept indices into the input, in descending-score order (int64).
"""
import numpy as np


def nms(boxes, scores, thr):
    boxes = np.asarray(boxes)
    scores = np.asarray(scores)
    n = boxes.shape[0]
    if n == 0:
        return np.empty((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores, kind="stable")  # stable, descending
    # np.argsort(-s) is stable-descending for finite s; equal scores keep input order.
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = boxes.dtype.type(thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.size == 0:
            continue
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(boxes.dtype.type(0), xx2 - xx1)
        h = np.maximum(boxes.dtype.type(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)
