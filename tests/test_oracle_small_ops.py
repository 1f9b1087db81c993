"""CPU: the oracle restatements vs golden vectors produced by the reference's own source
(oracle/tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import criterion as ocrit
from oracle import decode as odec
from oracle import nms as onms
from oracle import targets as otgt


def _case_noise(g, tag, ngv):
    np.random.seed(int(g[f"{tag}_seed"]))
    return np.random.rand(63, 63, 25, ngv) if ngv else np.zeros((63, 63, 25, 0))


def _valid(boxes):
    return boxes[~np.logical_or(boxes[:, 2] <= boxes[:, 0], boxes[:, 3] <= boxes[:, 1])]


@pytest.mark.parametrize("ci", range(5))
def test_targets_match_reference(golden, ci):
    g = golden("targets")
    t = g["templates"]
    tag = f"c{ci}"
    boxes = g[f"{tag}_boxes"]
    bv = _valid(boxes)
    iou = otgt.dense_overlap(-1, -1, 8, 8, 63, 63, t[:, 0], t[:, 1], t[:, 2], t[:, 3],
                             bv[:, 0], bv[:, 1], bv[:, 2], bv[:, 3])
    full = bool(g[f"{tag}_full"])
    sub = (lambda a: a) if full else (lambda a: a[::4, ::4])
    assert np.array_equal(sub(iou), g[f"{tag}_iou"])                  # bit-exact, float64
    assert iou.sum() == float(g[f"{tag}_iou_sum"])
    pad = otgt.get_padding(t, g[f"{tag}_paste"])
    if int(g[f"{tag}_flip"]):
        pad = np.fliplr(pad)
    assert np.array_equal(np.packbits(pad), g[f"{tag}_pad"])
    cm, rm, ioup = otgt.get_heatmaps(boxes.copy(), t, pad, noise=_case_noise(g, tag, bv.shape[0]))
    assert np.array_equal(cm.astype(np.int8), g[f"{tag}_class"])
    assert np.array_equal(sub(rm), g[f"{tag}_reg"])
    assert rm.sum() == float(g[f"{tag}_reg_sum"])
    if bv.shape[0]:
        assert np.array_equal(ioup.max(axis=(0, 1, 2)), g[f"{tag}_ioupert_max"])


def test_dense_overlap_loop_equals_vectorised(golden):
    t = golden("targets")["templates"][[0, 7, 24]]
    rng = np.random.RandomState(3)
    b = np.array([[10.2, 20.1, 60.7, 90.3], [-5.0, -3.0, 8.0, 9.0], [30.0, 30.0, 30.5, 31.0]])
    a = otgt.dense_overlap_loop(-1, -1, 8, 8, 9, 7, t[:, 0], t[:, 1], t[:, 2], t[:, 3], b[:, 0], b[:, 1], b[:, 2], b[:, 3])
    v = otgt.dense_overlap(-1, -1, 8, 8, 9, 7, t[:, 0], t[:, 1], t[:, 2], t[:, 3], b[:, 0], b[:, 1], b[:, 2], b[:, 3])
    assert np.array_equal(a, v)


@pytest.mark.parametrize("ci", range(4))
def test_decode_matches_reference(golden, ci):
    g = golden("decode")
    t = golden("targets")["templates"]
    tag = f"d{ci}"
    rf = otgt.RF
    b, s = odec.get_bboxes(g[f"{tag}_score_cls"], g[f"{tag}_score_reg"], g[f"{tag}_prob"].copy(), t,
                           float(g[f"{tag}_thr"]), rf, float(g[f"{tag}_scale"]))
    assert np.array_equal(b, g[f"{tag}_boxes"])
    assert np.array_equal(s, g[f"{tag}_scores"])
    assert b.shape[0] > 0 or ci == 0


def test_decode_w_lt_25_raises_like_reference(golden):
    g = golden("decode")
    assert int(g["w_lt_25_raises_indexerror"]) == 1
    z = np.zeros((1, 5, 20, 25), np.float32)
    with pytest.raises(IndexError):
        odec.get_bboxes(z, np.zeros((1, 5, 20, 100), np.float32), z.copy(), golden("targets")["templates"],
                        0.5, otgt.RF, 1)
    # intended behaviour is available explicitly and does not raise
    odec.get_bboxes(z, np.zeros((1, 5, 20, 100), np.float32), z.copy(), golden("targets")["templates"],
                    0.5, otgt.RF, 1, mask_axis="template")


def test_invalid_template_ids_constant_set(golden):
    t = golden("targets")["templates"]
    for s in (0.25, 0.5, 1, 2):
        assert sorted(odec.invalid_template_ids(t, s).tolist()) == [0, 1, 2, 3] + list(range(12, 25))


@pytest.mark.parametrize("ci", range(4))
def test_criterion_matches_reference(golden, ci):
    g = golden("criterion")
    tag = f"k{ci}"
    out = torch.from_numpy(g[f"{tag}_output"])
    cm = torch.from_numpy(g[f"{tag}_class_map"].astype(np.float32))
    rm = torch.from_numpy(g[f"{tag}_reg_map"])
    np.random.seed(int(g[f"{tag}_seed"]))
    r = ocrit.criterion(out, cm, rm)
    ref = g[f"{tag}_loss"]
    assert np.allclose([r["total"], r["cls"], r["reg"]], ref, rtol=1e-6)
    assert np.array_equal(r["grad"].numpy(), g[f"{tag}_grad"])
    # balance sampling caps
    fin = r["class_map_final"].numpy()
    for b in range(fin.shape[0]):
        assert (fin[b] == 1).sum() <= 128 and (fin[b] == -1).sum() <= 128


@pytest.mark.parametrize("ci", range(5))
def test_nms_restated_vectors(golden, ci):
    g = golden("nms")
    tag = f"n{ci}"
    keep = onms.nms(g[f"{tag}_boxes"], g[f"{tag}_scores"], 0.3)
    assert np.array_equal(keep, g[f"{tag}_keep"])
    if ci == 3:
        assert 200 < keep.size < 1251


def test_nms_semantics_small():
    # strict '>' : IoU exactly 0.5 with thr 0.5 is NOT suppressed; stable order on ties
    boxes = np.array([[0, 0, 2, 1], [1, 0, 3, 1], [0, 0, 2, 1]], dtype=np.float64)   # IoU(0,1) = 1/3, IoU(0,2) = 1
    scores = np.array([1.0, 1.0, 1.0])
    assert onms.nms(boxes, scores, 0.3).tolist() == [0]
    assert onms.nms(boxes, scores, 1 / 3).tolist() == [0, 1]
    z = np.array([[5, 5, 5, 5], [5, 5, 5, 5]], dtype=np.float64)                      # 0/0 -> NaN -> kept
    assert onms.nms(z, np.array([0.1, 0.2]), 0.3).tolist() == [1, 0]
