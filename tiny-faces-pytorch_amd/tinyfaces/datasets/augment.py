"""Training-time augmentation of the reference (WIDERFace.process_inputs, tinyfaces/datasets/wider_face.py:133-192, and
DataProcessor.crop_image, tinyfaces/datasets/processor.py:41-112) with the image work on the GPU.

Division of labour: the random decisions (scale, crop origin, paste origin, flip) and the bounding-box bookkeeping are a few
dozen floats per image and stay on the host, drawn from np.random in the reference's order so that a seeded run makes the
same decisions; the pixels -- resize, crop, paste on the mean colour, flip, ToTensor + Normalize -- are ONE HIP kernel per
image (ops.image_prepare) that only resamples the 500x500 window instead of resizing the whole image on the CPU.
The output tensor is bit-identical to the reference's CPU pipeline followed by main.py:44-46's transforms."""
from copy import deepcopy

import numpy as np
import torch

from .. import ops


def rect_dist(I, J):
    """tinyfaces/metrics.py:44-74: 1 - IoU (with the +1 pixel convention), clipped to [0, 1]; undefined ratios count as IoU 0."""
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


def crop_decisions(img_h, img_w, bboxes, input_size=(500, 500), neg_thresh=0.3, rng=np.random):
    """The host half of DataProcessor.crop_image (processor.py:41-112): the crop window, where it is pasted, and the boxes
    that survive (shifted, clamped, filtered by overlap with their cropped part).  Same np.random draws, same order."""
    ih, iw = input_size
    max_crop_x = max(1, img_w - iw + 1)
    max_crop_y = max(1, img_h - ih + 1)
    crop_x1 = rng.randint(0, max_crop_x)
    crop_y1 = rng.randint(0, max_crop_y)
    crop_x2 = min(img_w, crop_x1 + iw)
    crop_y2 = min(img_h, crop_y1 + ih)
    crop_h, crop_w = crop_y2 - crop_y1, crop_x2 - crop_x1
    paste = [0, 0, 0, 0]
    paste[0] = rng.randint(0, iw - crop_w + 1)
    paste[1] = rng.randint(0, ih - crop_h + 1)
    paste[2], paste[3] = paste[0] + crop_w, paste[1] + crop_h
    bboxes = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
    if bboxes.shape[0] > 0:
        tbox = deepcopy(bboxes)
        tbox[:, 0] = np.maximum(tbox[:, 0], crop_x1)
        tbox[:, 1] = np.maximum(tbox[:, 1], crop_y1)
        tbox[:, 2] = np.minimum(tbox[:, 2], crop_x2)
        tbox[:, 3] = np.minimum(tbox[:, 3], crop_y2)
        overlap = 1 - rect_dist(tbox, bboxes)
        bboxes[:, 0] = bboxes[:, 0] - crop_x1 + paste[0]        # (b - c) + p, in the reference's order: the last bit matters
        bboxes[:, 1] = bboxes[:, 1] - crop_y1 + paste[1]
        bboxes[:, 2] = bboxes[:, 2] - crop_x1 + paste[0]
        bboxes[:, 3] = bboxes[:, 3] - crop_y1 + paste[1]
        bboxes[:, 0] = np.minimum(iw, np.maximum(0, bboxes[:, 0]))
        bboxes[:, 1] = np.minimum(ih, np.maximum(0, bboxes[:, 1]))
        bboxes[:, 2] = np.minimum(iw, np.maximum(1, bboxes[:, 2]))
        bboxes[:, 3] = np.minimum(ih, np.maximum(1, bboxes[:, 3]))
        invalid = (bboxes[:, 2] <= bboxes[:, 0]) | (bboxes[:, 3] <= bboxes[:, 1]) | (overlap < neg_thresh)
        bboxes = bboxes[~invalid]
    return (crop_y1, crop_x1, crop_h, crop_w), paste, bboxes


def process_inputs(image_u8, bboxes, input_size=(500, 500), neg_thresh=0.3, rng=np.random, out=None, mean=ops.IMAGE_MEAN,
                   std=ops.IMAGE_STD):
    """WIDERFace.process_inputs up to get_heatmaps, for a decoded uint8 (H, W, 3) image that is already on the device.
    Returns (x float32 (3, ih, iw) device tensor = Normalize(ToTensor(augmented image)), bboxes float64 (G', 4),
    paste_box [x1, y1, x2, y2], flip) -- paste_box / flip are what ops.dense_overlap_targets* take for the padding mask."""
    if not image_u8.is_cuda:
        raise RuntimeError("process_inputs: the image must be on the GPU (no CPU fallback for the hot path)")
    H, W = int(image_u8.shape[0]), int(image_u8.shape[1])
    bboxes = np.array(bboxes, dtype=np.float64).reshape(-1, 4)
    rnd = rng.rand()                                             # wider_face.py:135
    rh, rw = H, W
    if rnd < 1 / 3:
        rh, rw = int(0.5 * H), int(0.5 * W)
        bboxes = bboxes / 2
    elif rnd > 2 / 3:
        rh, rw = int(2 * H), int(2 * W)
        bboxes = bboxes * 2
    crop, paste, bboxes = crop_decisions(rh, rw, bboxes, input_size, neg_thresh, rng)
    flip = bool(rng.rand() > 0.5)                                # wider_face.py:155
    if flip:
        lx1, lx2 = np.array(bboxes[:, 0]), np.array(bboxes[:, 2])
        bboxes[:, 0] = input_size[1] - lx2 + 1
        bboxes[:, 2] = input_size[1] - lx1 + 1
    x = ops.image_prepare(image_u8, resized_hw=(rh, rw), crop=crop, paste=(paste[1], paste[0]), flip=flip, out_hw=input_size, mean=mean,
                          std=std, out=out)
    return x, bboxes, paste, flip


def collate_device(samples, templates_d, seed=0):
    """[(x, bboxes, paste_box, flip)] -> (img (B,3,H,W), class_map, regression_map) with the targets assigned on the GPU
    (what WIDERFace.__getitem__ + the default collate yield, wider_face.py:219-222)."""
    xs = torch.stack([s[0] for s in samples])
    cm, rm = ops.dense_overlap_targets([s[1] for s in samples], templates_d, paste_boxes=[s[2] for s in samples],
                                       flips=[int(s[3]) for s in samples], seed=seed, device=xs.device)      # boxes are already mirrored; flips mirrors the padding mask
    return xs, cm, rm
