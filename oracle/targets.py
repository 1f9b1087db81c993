"""Oracle: dense_overlap anchor-IoU + heat-map target assignment (numpy, float64).

Restates (test infrastructure only; see oracle/__init__.py):
  * compute_dense_overlap      tinyfaces/datasets/dense_overlap.py:4-75
  * DataProcessor.get_padding  tinyfaces/datasets/processor.py:114-150
  * DataProcessor.get_regression                      processor.py:152-211
  * DataProcessor.get_heatmaps                        processor.py:213-277
Pinned against golden vectors produced by the reference's own source
(tests/golden/targets_*.npz, oracle/tools/make_golden.py).
"""
import numpy as np

RF = {"size": [859, 859], "stride": [8, 8], "offset": [-1, -1]}  # wider_face.py:55
HEATMAP_SIZE = (63, 63)                                          # wider_face.py:25
POS_THRESH, NEG_THRESH = 0.7, 0.3                                # wider_face.py:26-27


def dense_overlap_loop(ofx, ofy, stx, sty, vsx, vsy, dx1, dy1, dx2, dy2, gx1, gy1, gx2, gy2):
    """Literal quad loop of dense_overlap.py:30-75 (zmx=zmy=1).  Small cases only."""
    nt, ng = dx1.shape[0], gx1.shape[0]
    ov = np.zeros((vsy, vsx, nt, ng))
    for i in range(ng):
        bw, bh = gx2[i] - gx1[i] + 1, gy2[i] - gy1[i] + 1
        barea = bw * bh
        for j in range(nt):
            fh = dy2[j] - dy1[j] + 1
            fw = dx2[j] - dx1[j] + 1
            farea = fw * fh
            for x in range(vsx):
                for y in range(vsy):
                    cx = ofx + x * (stx / 1)
                    cy = ofy + y * (sty / 1)
                    x1, y1, x2, y2 = dx1[j] + cx, dy1[j] + cy, dx2[j] + cx, dy2[j] + cy
                    iw = min(x2, gx2[i]) - max(x1, gx1[i]) + 1
                    ih = min(y2, gy2[i]) - max(y1, gy1[i]) + 1
                    if ih > 0 and iw > 0:
                        ia = iw * ih
                        ov[y, x, j, i] = ia / (farea + barea - ia)
    return np.around(ov, decimals=14)


def dense_overlap(ofx, ofy, stx, sty, vsx, vsy, dx1, dy1, dx2, dy2, gx1, gy1, gx2, gy2):
    """Vectorised, operation-for-operation float64 equivalent of dense_overlap.py:30-75.
    Every scalar expression is evaluated with the same operand order so the result is
    bit-identical to the loop (asserted in tests/test_oracle_targets.py)."""
    nt, ng = dx1.shape[0], gx1.shape[0]
    if ng == 0:
        return np.zeros((vsy, vsx, nt, 0))
    cx = (ofx + np.arange(vsx) * (stx / 1)).reshape(1, vsx, 1, 1)      # :50
    cy = (ofy + np.arange(vsy) * (sty / 1)).reshape(vsy, 1, 1, 1)      # :51
    d = lambda a: a.reshape(1, 1, nt, 1)
    g = lambda a: a.reshape(1, 1, 1, ng)
    x1, y1, x2, y2 = d(dx1) + cx, d(dy1) + cy, d(dx2) + cx, d(dy2) + cy  # :53-56
    xx1, yy1 = np.maximum(x1, g(gx1)), np.maximum(y1, g(gy1))             # :58-59
    xx2, yy2 = np.minimum(x2, g(gx2)), np.minimum(y2, g(gy2))             # :60-61
    int_w, int_h = xx2 - xx1 + 1, yy2 - yy1 + 1                           # :63-64
    bbox_area = g((gx2 - gx1 + 1) * (gy2 - gy1 + 1))                      # :33-34
    filter_area = d((dx2 - dx1 + 1) * (dy2 - dy1 + 1))                    # :39-42
    int_area = int_w * int_h
    union = filter_area + bbox_area - int_area                           # :68
    valid = (int_h > 0) & (int_w > 0)                                     # :66
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = np.where(valid, int_area / union, 0.0)
    return np.around(ov, decimals=14)                                     # :75


def get_padding(templates, paste_box, rf=RF, heatmap_size=HEATMAP_SIZE):
    """processor.py:114-150.  Returns bool (vsy, vsx, nt)."""
    ofy, ofx = rf["offset"]
    sty, stx = rf["stride"]
    vsy, vsx = heatmap_size
    coarse_x, coarse_y = np.meshgrid(ofx + np.arange(vsx) * stx, ofy + np.arange(vsy) * sty)
    dx1, dy1, dx2, dy2 = (templates[:, k] for k in range(4))
    xx1 = coarse_x[:, :, None] + dx1[None, None, :]
    yy1 = coarse_y[:, :, None] + dy1[None, None, :]
    xx2 = coarse_x[:, :, None] + dx2[None, None, :]
    yy2 = coarse_y[:, :, None] + dy2[None, None, :]
    return (xx1 < paste_box[0] + 1) | (yy1 < paste_box[1] + 1) | (xx2 > paste_box[2]) | (yy2 > paste_box[3])


def get_regression(bboxes, templates, iou, noise, rf=RF, heatmap_size=HEATMAP_SIZE):
    """processor.py:152-211 with the `np.random.rand(*iou.shape)` draw (:195) injected as
    `noise` (same shape as iou, values in [0,1))."""
    ofy, ofx = rf["offset"]
    sty, stx = rf["stride"]
    vsy, vsx = heatmap_size
    coarse_xx, coarse_yy = np.meshgrid(ofx + np.arange(vsx) * stx, ofy + np.arange(vsy) * sty)
    dx1, dy1, dx2, dy2 = (templates[:, k] for k in range(4))
    ng = bboxes.shape[0]
    fxx1, fyy1, fxx2, fyy2 = (bboxes[:, k].reshape(1, 1, 1, ng) for k in range(4))
    dhh = (dy2 - dy1 + 1).reshape(1, 1, -1, 1)
    dww = (dx2 - dx1 + 1).reshape(1, 1, -1, 1)
    fcx, fcy = (fxx1 + fxx2) / 2, (fyy1 + fyy2) / 2
    tx = np.divide(fcx - coarse_xx.reshape(vsy, vsx, 1, 1), dww)
    ty = np.divide(fcy - coarse_yy.reshape(vsy, vsx, 1, 1), dhh)
    fhh, fww = fyy2 - fyy1 + 1, fxx2 - fxx1 + 1
    tw = np.log(np.divide(fww, dww))
    th = np.log(np.divide(fhh, dhh))
    iou = iou + (1e-6 * noise)                                            # :195
    best = iou.argmax(axis=3)                                             # :197
    i0, i1, i2 = np.indices(iou.shape[:-1])
    tx, ty = tx[i0, i1, i2, best], ty[i0, i1, i2, best]
    tw = np.broadcast_to(tw, iou.shape)[i0, i1, i2, best]
    th = np.broadcast_to(th, iou.shape)[i0, i1, i2, best]
    return np.concatenate((tx, ty, tw, th), axis=2), iou


def get_heatmaps(bboxes, templates, pad_mask, noise=None, rf=RF, heatmap_size=HEATMAP_SIZE,
                 pos_thresh=POS_THRESH, neg_thresh=NEG_THRESH, overlap_fn=dense_overlap):
    """processor.py:213-277.  `noise`: (vsy, vsx, nt, ng_valid) uniform [0,1) draws, or None
    (then drawn from np.random like the reference).  Returns (class_maps (vsy,vsx,nt),
    regress_maps (vsy,vsx,4nt), perturbed iou)."""
    ofy, ofx = rf["offset"]
    sty, stx = rf["stride"]
    vsy, vsx = heatmap_size
    nt = templates.shape[0]
    class_maps = -np.ones((vsy, vsx, nt))
    regress_maps = np.zeros((vsy, vsx, nt * 4))
    dx1, dx2 = templates[:, 0], templates[:, 2]
    dy1, dy2 = templates[:, 1], templates[:, 3]
    invalid = np.logical_or(bboxes[:, 2] <= bboxes[:, 0], bboxes[:, 3] <= bboxes[:, 1])   # :228-230
    bboxes = np.delete(bboxes, np.where(invalid), axis=0)
    ng = bboxes.shape[0]
    iou = np.zeros((vsy, vsx, nt, ng))
    if ng > 0:
        iou = overlap_fn(ofx, ofy, stx, sty, vsx, vsy, dx1, dy1, dx2, dy2,
                         bboxes[:, 0], bboxes[:, 1], bboxes[:, 2], bboxes[:, 3])
        if noise is None:
            noise = np.random.rand(*iou.shape)
        regress_maps, iou = get_regression(bboxes, templates, iou, noise, rf, heatmap_size)
        best_iou = iou.max(axis=3)
        per_object = np.reshape(iou, (-1, ng))                            # :252
        fbest = np.argmax(per_object, axis=0)
        iou_ = np.amax(per_object, axis=0)
        fbest = np.unravel_index(fbest[iou_ > neg_thresh], iou.shape[:-1])
        class_maps[fbest] = 1                                             # :257
        class_maps = np.maximum(class_maps, (best_iou >= pos_thresh) * 2 - 1)   # :260
        gray = -np.ones(class_maps.shape)
        gray[np.bitwise_and(neg_thresh <= best_iou, best_iou < pos_thresh)] = 0
        class_maps = np.maximum(class_maps, gray)                         # :269
    non_neg_border = np.bitwise_and(pad_mask, class_maps != -1)           # :272
    class_maps[non_neg_border] = 0
    regress_maps[:, :, :nt][non_neg_border] = 0                           # :274 (tx block only: D5)
    return class_maps, regress_maps, iou
