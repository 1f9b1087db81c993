"""Synthetic 500x500 crops + random boxes (BASELINE.json configs[2]) and
the device-side target assigner that replaces DataProcessor.get_heatmaps in the loader."""
import numpy as np
import torch
from torch.utils import data

from .. import ops

INPUT_SIZE = (500, 500)          # tinyfaces/datasets/wider_face.py:24
HEATMAP_SIZE = (63, 63)          # :25
POS_THRESH, NEG_THRESH = 0.7, 0.3   # :26-27


def random_boxes(rng, size=500):
    """G ~ U{1..16}; width log-uniform 8-200 px; h/w ~ U[1,1.5]; placed inside the crop."""
    g = int(rng.randint(1, 17))
    w = np.exp(rng.uniform(np.log(8), np.log(200), g))
    h = np.minimum(w * rng.uniform(1.0, 1.5, g), size - 2)
    w = np.minimum(w, size - 2)
    x1 = rng.uniform(1, size - w)
    y1 = rng.uniform(1, size - h)
    return np.stack([x1, y1, x1 + w, y1 + h], 1)


class TargetAssigner:
    """boxes (+ paste box / flip) -> (class_map, regression_map) on the GPU.
    Replaces DataProcessor.get_padding/get_heatmaps (tinyfaces/datasets/processor.py:114-277)."""

    def __init__(self, templates, heatmap_size=HEATMAP_SIZE, rf=ops.RF, pos_thresh=POS_THRESH, neg_thresh=NEG_THRESH, seed=0):
        self.templates, self.heatmap_size, self.rf = templates, heatmap_size, rf
        self.pos_thresh, self.neg_thresh, self.seed, self.calls = pos_thresh, neg_thresh, seed, 0

    def __call__(self, boxes_per_image, paste_boxes=None, flips=None, device="cuda", noise=None):
        self.calls += 1
        return ops.dense_overlap_targets(boxes_per_image, self.templates, self.heatmap_size, self.rf, paste_boxes, flips,
                                         noise=noise, seed=self.seed * 1000003 + self.calls, pos_thresh=self.pos_thresh,
                                         neg_thresh=self.neg_thresh, device=device)


class SyntheticCrops(data.Dataset):
    """train=True: (img, boxes) -> collate -> (img, class_map, regression_map) batches like WIDERFace's training samples
    (wider_face.py:219-222).  train=False: the val contract of wider_face.py:224-233 -- an image tensor in [0, 1] of
    `val_size` and its (made-up) relative path; collate adds the batch axis / list exactly like torch's default collate."""

    def __init__(self, templates, length=256, seed=0, train=True, device="cuda", val_size=(960, 1280), img_transforms=None):
        self.templates, self.length, self.seed, self.train = templates, length, seed, train
        self.assigner = TargetAssigner(templates, seed=seed)
        self.device = device
        self.rf = ops.RF
        self.transforms = img_transforms
        self.val_size = val_size

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        rng = np.random.RandomState(self.seed * 7919 + i)
        g = torch.Generator().manual_seed(self.seed * 7919 + i)
        if not self.train:
            return torch.rand(3, *self.val_size, generator=g), f"synthetic/img_{i}.jpg"
        img = torch.randn(3, *INPUT_SIZE, generator=g)
        return img, random_boxes(rng)

    def collate(self, batch):
        if not self.train:
            return torch.stack([b[0] for b in batch]), [b[1] for b in batch]
        imgs = torch.stack([b[0] for b in batch])
        boxes = [b[1] for b in batch]
        if not torch.cuda.is_available():
            raise RuntimeError("SyntheticCrops: target assignment runs on the GPU only (no CPU fallback)")
        cm, rm = self.assigner(boxes, paste_boxes=[[0, 0, 500, 500]] * len(batch), device=self.device)
        return imgs, cm, rm


# ---- r6: a data set the detector can actually LEARN: a fixed list of images with pasted "faces" -------------------------------
# SyntheticCrops is noise with random boxes: right for throughput, nothing to learn.  Here every image is a dim noise background with a few
# copies of ONE fixed high-contrast pattern (a 12 x 10 cell "face": bright oval-ish field, two dark eye cells, a dark mouth bar) resized to
# log-uniform widths; the ground-truth box is the pasted rectangle.  The SAME uint8 pixels serve training (ToTensor + Normalize) and evaluation
# (the val contract: a float tensor in [0, 1] + a relative path), so `main.py synthetic-faces ...` -> checkpoint -> `evaluate_model.py
# synthetic-faces` -> write_results -> wider_eval.evaluate_setting closes the loop of the reference's Makefile:11-21 without WIDER assets.
_FACE = np.array([[.55, .8, .85, .9, .9, .9, .9, .85, .8, .55],
                  [.8, .95, .95, .95, .95, .95, .95, .95, .95, .8],
                  [.85, .95, .95, .95, .95, .95, .95, .95, .95, .85],
                  [.9, .2, .1, .2, .95, .95, .2, .1, .2, .9],
                  [.9, .2, .05, .2, .95, .95, .2, .05, .2, .9],
                  [.9, .95, .95, .95, .7, .7, .95, .95, .95, .9],
                  [.9, .95, .95, .95, .6, .6, .95, .95, .95, .9],
                  [.9, .95, .95, .8, .55, .55, .8, .95, .95, .9],
                  [.85, .95, .3, .15, .15, .15, .15, .3, .95, .85],
                  [.85, .95, .6, .3, .25, .25, .3, .6, .95, .85],
                  [.8, .95, .95, .95, .95, .95, .95, .95, .95, .8],
                  [.55, .8, .85, .9, .9, .9, .9, .85, .8, .55]], np.float32)


def face_image(rng, size=(500, 500), n_faces=(2, 5), widths=(28, 140)):
    """(uint8 HxWx3 image, boxes (G,4) x1 y1 x2 y2): non-overlapping faces of log-uniform width, aspect 1.2, on a dim noise background."""
    H, W = size
    img = (rng.rand(H, W, 3) * 60 + 30).astype(np.float32)
    boxes = []
    for _ in range(int(rng.randint(n_faces[0], n_faces[1] + 1))):
        for _try in range(50):
            w = float(np.exp(rng.uniform(np.log(widths[0]), np.log(widths[1]))))
            w, h = int(round(w)), int(round(w * 1.2))
            x1, y1 = int(rng.randint(2, W - w - 2)), int(rng.randint(2, H - h - 2))
            b = np.array([x1, y1, x1 + w, y1 + h], np.float64)
            if all(min(b[2], o[2]) - max(b[0], o[0]) < -4 or min(b[3], o[3]) - max(b[1], o[1]) < -4 for o in boxes):
                break
        else:
            continue
        yy = np.minimum((np.arange(h) * _FACE.shape[0]) // h, _FACE.shape[0] - 1)
        xx = np.minimum((np.arange(w) * _FACE.shape[1]) // w, _FACE.shape[1] - 1)
        patch = _FACE[yy][:, xx]
        tint = 0.85 + 0.15 * rng.rand(3).astype(np.float32)
        img[y1:y1 + h, x1:x1 + w] = patch[:, :, None] * 255.0 * tint
        boxes.append(b)
    return np.clip(img, 0, 255).astype(np.uint8), np.stack(boxes) if boxes else np.zeros((0, 4))


class SyntheticFaces(data.Dataset):
    """`length` fixed images (seeded): train=True -> (normalised img, boxes) -> collate -> (img, class_map, regression_map);
    train=False -> (float tensor in [0, 1], "faces/img_<i>.jpg") like the val contract of wider_face.py:224-233."""
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    N_IMAGES = 8          # distinct images; a longer data set walks them again (sample i = image i % N_IMAGES): `--synthetic-len` = samples per epoch
    VAL_ZOOM = 2          # evaluation sees every image at twice the size (pixel replication, boxes x 2): the reference's pyramid goes down to 1/4 and its
                          # template mask indexes the heat-map width with the template index (defect D1: IndexError below 25 columns = 800 px at 1/4)

    def __init__(self, templates, length=8, seed=0, train=True, device="cuda", size=(500, 500), img_transforms=None):
        self.templates, self.length, self.seed, self.train, self.device, self.size = templates, length, seed, train, device, size
        self.assigner = TargetAssigner(templates, seed=seed)
        self.rf, self.transforms = ops.RF, img_transforms
        self.samples = [face_image(np.random.RandomState(seed * 7919 + 13 * i + 1), size) for i in range(min(length, self.N_IMAGES))]

    def __len__(self):
        return self.length

    def ground_truth(self):
        """{image name: (G,4) boxes x, y, w, h} in the inclusive-pixel convention write_results uses."""
        z = self.VAL_ZOOM
        return {f"img_{i}": np.column_stack([z * b[:, 0], z * b[:, 1], z * (b[:, 2] - b[:, 0]) + 1, z * (b[:, 3] - b[:, 1]) + 1]) for i, (_, b) in enumerate(self.samples)}

    def __getitem__(self, i):
        i = i % len(self.samples)
        u8, boxes = self.samples[i]
        if not self.train:
            z = self.VAL_ZOOM
            return torch.from_numpy(np.repeat(np.repeat(u8, z, axis=0), z, axis=1)).permute(2, 0, 1).float().div(255), f"faces/img_{i}.jpg"
        x = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
        mean, std = torch.tensor(self.MEAN).view(3, 1, 1), torch.tensor(self.STD).view(3, 1, 1)
        return (x - mean) / std, boxes

    def collate(self, batch):
        if not self.train:
            return torch.stack([b[0] for b in batch]), [b[1] for b in batch]
        imgs = torch.stack([b[0] for b in batch])
        H, W = self.size
        cm, rm = self.assigner([b[1] for b in batch], paste_boxes=[[0, 0, W, H]] * len(batch), device=self.device)
        return imgs, cm, rm
