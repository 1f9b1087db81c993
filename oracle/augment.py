"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the training-time image pipeline that sits in
front of the hot path and of the PIL resize both pipelines rely on:

  * pil_resize_u8        Pillow's two-pass 8-bit resample with the BILINEAR filter (third party: Pillow 12.2,
                         src/libImaging/Resample.c -- precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/
                         Vertical_8bpc); reached from tinyfaces/datasets/wider_face.py:139,145 and tinyfaces/evaluation.py:46
                         through torchvision.transforms.functional.resize.  Pinned against PIL itself (tests/test_oracle_augment.py).
  * crop_image           tinyfaces/datasets/processor.py:41-112
  * process_inputs       tinyfaces/datasets/wider_face.py:133-192 (without get_heatmaps, which oracle/targets.py restates)
  * to_normalized_tensor main.py:44-46 (ToTensor + Normalize)

np.random is drawn in the reference's order: rand() [scale], randint x4 [crop x, crop y, paste x, paste y], rand() [flip]."""
import math
from copy import deepcopy

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size):
    """precompute_coeffs (in0 = 0, in1 = in_size) + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0).
    Returns (xmin[out], count[out], k[out][ksize] int32 fixed point)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        if lo < 0:
            lo = 0
        hi = int(center + support + 0.5)
        if hi > in_size:
            hi = in_size
        n = hi - lo
        w = [_bilinear((x + lo - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_u8(img, out_h, out_w):
    """img uint8 (H, W, C) -> uint8 (out_h, out_w, C), == PIL.Image.resize((out_w, out_h), BILINEAR).
    Horizontal pass first (only when the width changes), then vertical (only when the height changes), uint8 in between."""
    img = np.asarray(img, dtype=np.uint8)
    H, W, C = img.shape
    cur = img
    if out_w != W:
        xmin, cnt, kk = resample_coeffs(W, out_w)
        acc = np.full((H, out_w, C), 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(kk.shape[1]):
            use = t < cnt
            src = np.minimum(xmin + t, W - 1)
            acc += np.where(use[None, :, None], cur[:, src, :].astype(np.int64) * kk[:, t][None, :, None], 0)
        cur = _clip8(acc)
    if out_h != H:
        ymin, cnt, kk = resample_coeffs(H, out_h)
        acc = np.full((out_h, cur.shape[1], C), 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(kk.shape[1]):
            use = t < cnt
            src = np.minimum(ymin + t, H - 1)
            acc += np.where(use[:, None, None], cur[src, :, :].astype(np.int64) * kk[:, t][:, None, None], 0)
        cur = _clip8(acc)
    return cur


def rect_dist(I, J):
    """tinyfaces/metrics.py:44-74 (1 - IoU with the +1 pixel convention, clipped to [0, 1]; 0/0 -> 1)."""
    I, J = np.atleast_2d(I).astype(np.float64), np.atleast_2d(J).astype(np.float64)
    aI = (I[:, 2] - I[:, 0] + 1) * (I[:, 3] - I[:, 1] + 1)
    aJ = (J[:, 2] - J[:, 0] + 1) * (J[:, 3] - J[:, 1] + 1)
    x1, y1 = np.maximum(I[:, 0], J[:, 0]), np.maximum(I[:, 1], J[:, 1])
    x2, y2 = np.minimum(I[:, 2], J[:, 2]), np.minimum(I[:, 3], J[:, 3])
    aIJ = (x2 - x1 + 1) * (y2 - y1 + 1) * np.logical_and(x2 > x1, y2 > y1)
    with np.errstate(all="ignore"):
        iou = aIJ / (aI + aJ - aIJ)
    iou[~np.isfinite(iou)] = 0
    return np.maximum(0.0, np.minimum(1.0, 1 - iou))


def crop_image(img, bboxes, input_size=(500, 500), img_means=(0.485, 0.456, 0.406), neg_thresh=0.3, rng=np.random):
    """processor.py:41-112.  Returns (img_buf int8 (ih, iw, 3), bboxes, paste_box [x1, y1, x2, y2])."""
    ih, iw = input_size
    max_crop_x = max(1, img.shape[1] - iw + 1)
    max_crop_y = max(1, img.shape[0] - ih + 1)
    crop_x1 = rng.randint(0, max_crop_x)
    crop_y1 = rng.randint(0, max_crop_y)
    crop_x2 = min(img.shape[1], crop_x1 + iw)
    crop_y2 = min(img.shape[0], crop_y1 + ih)
    crop_h, crop_w = crop_y2 - crop_y1, crop_x2 - crop_x1
    paste = [0, 0, 0, 0]
    paste[0] = rng.randint(0, iw - crop_w + 1)
    paste[1] = rng.randint(0, ih - crop_h + 1)
    paste[2], paste[3] = paste[0] + crop_w, paste[1] + crop_h
    buf = np.zeros((ih, iw, 3))
    for i, c in enumerate(img_means):
        buf[:, :, i] += c
    buf = (buf * 255).astype(np.int8)                      # 123, 116, 103: the mean colour, truncated
    buf[paste[1]:paste[3], paste[0]:paste[2], :] = img[crop_y1:crop_y2, crop_x1:crop_x2, :]    # uint8 -> int8 wraps; undone later
    bboxes = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
    if bboxes.shape[0] > 0:
        tbox = deepcopy(bboxes)
        tbox[:, 0] = np.maximum(tbox[:, 0], crop_x1)
        tbox[:, 1] = np.maximum(tbox[:, 1], crop_y1)
        tbox[:, 2] = np.minimum(tbox[:, 2], crop_x2)
        tbox[:, 3] = np.minimum(tbox[:, 3], crop_y2)
        overlap = 1 - rect_dist(tbox, bboxes)
        bboxes[:, 0] = bboxes[:, 0] - crop_x1 + paste[0]
        bboxes[:, 1] = bboxes[:, 1] - crop_y1 + paste[1]
        bboxes[:, 2] = bboxes[:, 2] - crop_x1 + paste[0]
        bboxes[:, 3] = bboxes[:, 3] - crop_y1 + paste[1]
        bboxes[:, 0] = np.minimum(iw, np.maximum(0, bboxes[:, 0]))
        bboxes[:, 1] = np.minimum(ih, np.maximum(0, bboxes[:, 1]))
        bboxes[:, 2] = np.minimum(iw, np.maximum(1, bboxes[:, 2]))
        bboxes[:, 3] = np.minimum(ih, np.maximum(1, bboxes[:, 3]))
        invalid = (bboxes[:, 2] <= bboxes[:, 0]) | (bboxes[:, 3] <= bboxes[:, 1]) | (overlap < neg_thresh)
        bboxes = bboxes[~invalid]
    return buf, bboxes, paste, (crop_x1, crop_y1, crop_w, crop_h)


def process_inputs(image_u8, bboxes, input_size=(500, 500), rng=np.random, **kw):
    """wider_face.py:133-192 up to (not including) get_heatmaps.  image_u8 (H, W, 3) uint8.
    Returns dict(img uint8 (ih, iw, 3), bboxes, paste_box, flip, scale, resized_hw, crop)."""
    H, W = image_u8.shape[:2]
    bboxes = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
    rnd = rng.rand()
    scale, rh, rw = 1, H, W
    if rnd < 1 / 3:
        scale, rh, rw = 0.5, int(0.5 * H), int(0.5 * W)
        bboxes = bboxes / 2
    elif rnd > 2 / 3:
        scale, rh, rw = 2, int(2 * H), int(2 * W)
        bboxes = bboxes * 2
    img = pil_resize_u8(image_u8, rh, rw) if scale != 1 else image_u8
    buf, bboxes, paste, crop = crop_image(img, bboxes, input_size, rng=rng, **kw)
    flip = rng.rand() > 0.5
    if flip:
        buf = np.fliplr(buf).copy()
        lx1, lx2 = np.array(bboxes[:, 0]), np.array(bboxes[:, 2])
        bboxes[:, 0] = input_size[1] - lx2 + 1
        bboxes[:, 2] = input_size[1] - lx1 + 1
    return dict(img=buf.astype(np.uint8), bboxes=bboxes, paste_box=paste, flip=bool(flip), scale=scale, resized_hw=(rh, rw), crop=crop)


def to_normalized_tensor(img_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """main.py:44-46: ToTensor (uint8 HWC -> float32 CHW / 255) then Normalize ((x - mean) / std), all in float32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)
