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
