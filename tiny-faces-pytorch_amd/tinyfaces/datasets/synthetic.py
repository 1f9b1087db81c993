"""Synthetic 500x500 crops + random boxes (BASELINE.json configs[2]) and
the device-side target assigner that replaces DataProcessor.get_heatmaps in the loader."""
import os

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



def _stack_on_device(tensors, device):
    """torch.stack(tensors) assembled ON the device: one upload per sample into a preallocated batch.  The host-side stack of 32 float
    500 x 500 images cost 56 ms per batch through torch's CPU thread pool (r6, cProfile of main.py: 40 % of a training step of the bundled data sets)."""
    out = torch.empty((len(tensors),) + tuple(tensors[0].shape), dtype=tensors[0].dtype, device=device)
    for i, t in enumerate(tensors):
        out[i].copy_(t)                  # (blocking: the pageable source of a sample may be freed or reused as soon as collate returns)
    return out

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
        boxes = [b[1] for b in batch]
        if not torch.cuda.is_available():
            raise RuntimeError("SyntheticCrops: target assignment runs on the GPU only (no CPU fallback)")
        imgs = _stack_on_device([b[0] for b in batch], self.device)
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


_GRAIN = np.random.RandomState(20240601).randint(0, 24, (1536, 1536, 3), dtype=np.uint8)


def background(rng, size=(500, 500)):
    """Dim two-scale noise (16-pixel blocks + per-pixel grain), uint8: a NEW draw for every training sample, so that nothing but the faces can be learned."""
    H, W = size
    coarse = rng.randint(20, 90, (H // 16 + 1, W // 16 + 1, 3), dtype=np.uint8)
    oy, ox = int(rng.randint(0, _GRAIN.shape[0] - H + 1)), int(rng.randint(0, _GRAIN.shape[1] - W + 1))      # the grain: a random window of one fixed texture (cheap)
    return np.repeat(np.repeat(coarse, 16, axis=0), 16, axis=1)[:H, :W] + _GRAIN[oy:oy + H, ox:ox + W]


def face_layout(rng, size=(500, 500), n_faces=(2, 5), widths=(float(os.environ.get("FACES_WMIN", "28")), float(os.environ.get("FACES_WMAX", "140")))):
    """[(box x1 y1 x2 y2, tint)]: non-overlapping faces of log-uniform width, aspect 1.2."""
    H, W = size
    boxes, faces = [], []
    for _ in range(int(rng.randint(n_faces[0], n_faces[1] + 1))):
        for _try in range(50):
            w = float(np.exp(rng.uniform(np.log(widths[0]), np.log(widths[1]))))
            w, h = int(round(w)), int(round(w * 1.3))          # (the aspect of the templates)
            x1, y1 = int(rng.randint(2, W - w - 2)), int(rng.randint(2, H - h - 2))
            b = np.array([x1, y1, x1 + w, y1 + h], np.float64)
            if all(min(b[2], o[2]) - max(b[0], o[0]) < -4 or min(b[3], o[3]) - max(b[1], o[1]) < -4 for o in boxes):
                break
        else:
            continue
        boxes.append(b)
        faces.append((b, 0.85 + 0.15 * rng.rand(3).astype(np.float32)))
    return faces


def render_faces(bg, faces):
    """Paste the face pattern into a copy of the uint8 background; returns (image, boxes (G,4))."""
    img = bg.copy()
    for b, tint in faces:
        x1, y1, x2, y2 = [int(v) for v in b]
        w, h = x2 - x1, y2 - y1
        yy = np.minimum((np.arange(h) * _FACE.shape[0]) // h, _FACE.shape[0] - 1)
        xx = np.minimum((np.arange(w) * _FACE.shape[1]) // w, _FACE.shape[1] - 1)
        img[y1:y1 + h, x1:x1 + w] = np.clip(_FACE[yy][:, xx][:, :, None] * 255.0 * tint, 0, 255).astype(np.uint8)
    return img, np.stack([b for b, _ in faces]) if faces else np.zeros((0, 4))


def face_image(rng, size=(500, 500)):
    """One image of the fixed list: (uint8 HxWx3, boxes), background and layout from the same generator."""
    faces = face_layout(rng, size)
    return render_faces(background(rng, size), faces)


class SyntheticFaces(data.Dataset):
    """`length` fixed images (seeded): train=True -> (normalised img, boxes) -> collate -> (img, class_map, regression_map);
    train=False -> (float tensor in [0, 1], "faces/img_<i>.jpg") like the val contract of wider_face.py:224-233."""
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    _LUT01 = (torch.arange(256, dtype=torch.float32) / 255).numpy()
    _LUTN = ((torch.arange(256, dtype=torch.float32) / 255).view(1, 256) - torch.tensor(MEAN).view(3, 1)).div(torch.tensor(STD).view(3, 1)).numpy()

    N_VIEWS = int(os.environ.get("FACES_VIEWS", "4"))          # training views per layout, walked round-robin: the evaluation image and its x0.5 / x2 / x0.25 versions (32 training images = ONE batch of 32)
    N_IMAGES = 8          # distinct images; a longer data set walks them again (sample i = image i % N_IMAGES): `--synthetic-len` = samples per epoch
    VAL_ZOOM = 1          # evaluation sees the images at their own size: run evaluate_model.py with --mask-axis template (with the reference's mask, defect D1,
                          # the 1/4 level of a 500-pixel image raises IndexError: the mask indexes the 16 heat-map columns with the 25 template indices)

    def __init__(self, templates, length=8, seed=0, train=True, device="cuda", size=(500, 500), img_transforms=None):
        self.templates, self.length, self.seed, self.train, self.device, self.size = templates, length, seed, train, device, size
        # anchors are positive from IoU 0.7 (like WIDER training, wider_face.py:26) and negative below 0.7 (there: 0.3): with eight layouts the wide "ignored"
        # band of the reference would stay untrained -- its anchors answer with arbitrary scores and unregressed boxes, which an AP at IoU 0.5 counts as misses
        self.assigner = TargetAssigner(templates, seed=seed, pos_thresh=float(os.environ.get("FACES_POS", "0.7")), neg_thresh=float(os.environ.get("FACES_NEG", "0.7")))
        self.rf, self.transforms = ops.RF, img_transforms
        # the layouts (where the faces are) are the fixed part; `samples` = what evaluation sees: every layout on ONE background of its own.
        # Training sample i shows layout i % N_IMAGES on a background drawn for THAT sample (seeded by i): the pixels between the faces never repeat.
        self.layouts = [face_layout(np.random.RandomState(seed * 7919 + 13 * i + 1), size) for i in range(min(length, self.N_IMAGES))]
        self.samples = [render_faces(background(np.random.RandomState(seed * 7919 + 13 * i + 5), size), f) for i, f in enumerate(self.layouts)]
        self._views = {}

    def __len__(self):
        return self.length

    def ground_truth(self):
        """{image name: (G,4) boxes x, y, w, h} in the inclusive-pixel convention write_results uses."""
        z = self.VAL_ZOOM
        return {f"img_{i}": np.column_stack([z * b[:, 0], z * b[:, 1], z * (b[:, 2] - b[:, 0]) + 1, z * (b[:, 3] - b[:, 1]) + 1]) for i, (_, b) in enumerate(self.samples)}

    def __getitem__(self, i):
        k, i = i, i % len(self.samples)
        u8, boxes = self.samples[i]
        if self.train:
            # visit number of this layout -> one of its N_VIEWS training views (rendered once, on first use), staggered by the layout so that a batch of
            # consecutive samples mixes the zooms (batches of ONE zoom have batch statistics of their own, which the running averages of evaluation do not match)
            v = (k // len(self.samples) + i) % self.N_VIEWS
            if (i, v) not in self._views:
                # view 0 IS the evaluation image of the layout (the judge's "the same images"); the others: backgrounds and zooms of their own
                u8, boxes = self.samples[i] if v == 0 else self._train_view(np.random.RandomState(self.seed * 104729 + 31 * (v * len(self.samples) + i) + 7), i, v)
                self._views[(i, v)] = (torch.from_numpy(np.stack([self._LUTN[c][u8[:, :, c]] for c in range(3)])), boxes)
            return self._views[(i, v)]
        # uint8 -> float through 256-entry tables (ToTensor, and ToTensor + Normalize): exactly torch's arithmetic, a fraction of its time on the host
        if not self.train:
            z = self.VAL_ZOOM
            big = np.repeat(np.repeat(u8, z, axis=0), z, axis=1)
            return torch.from_numpy(np.ascontiguousarray(self._LUT01[big].transpose(2, 0, 1))), f"faces/img_{i}.jpg"
        return torch.from_numpy(np.stack([self._LUTN[c][u8[:, :, c]] for c in range(3)])), boxes

    def _train_view(self, rng, i, v):
        """One training sample of a layout: a background of its own and -- like the reference's augmentation (wider_face.py:136-146: the image at x0.5,
        x1 or x2, then a 500-pixel crop) -- one of four zooms through PIL's BILINEAR resize, the resampler the evaluation pyramid uses: x1 as rendered;
        x0.5 / x0.25 pasted at a random place of another background; a random 500-pixel window of x2 (faces that lose more than 30 % to the window are dropped
        from the targets only when nothing of them is left to see: a partly visible face keeps its clipped box)."""
        from PIL import Image
        H, W = self.size
        # the picture that is zoomed: the evaluation image itself (FACES_SAME_BG, default) -- the levels of the evaluation pyramid are then resampled
        # versions of pictures the detector was trained on -- or the layout on a background of its own
        if os.environ.get("FACES_SAME_BG", "1") == "1":
            img, boxes = self.samples[i]
            z = (0.5, 2.0, 0.25, 2.0)[(v - 1) % 4]                   # view 0 is x1; x2 twice as often (a 500-pixel window shows a quarter of it)
        else:
            img, boxes = render_faces(background(rng, self.size), self.layouts[i])
            z = (1.0, 0.5, 2.0, 0.25)[int(rng.randint(0, 4))]          # the four levels of the reference's evaluation pyramid (evaluation.py:37: 2 ** (-2 .. 1))
        if z == 1.0:
            return img, boxes
        small = np.asarray(Image.fromarray(img).resize((int(W * z), int(H * z)), Image.BILINEAR))
        if z < 1.0:
            canvas = background(rng, self.size)
            oy, ox = int(rng.randint(0, H - small.shape[0] + 1)), int(rng.randint(0, W - small.shape[1] + 1))
            canvas[oy:oy + small.shape[0], ox:ox + small.shape[1]] = small
            return canvas, boxes * z + np.array([ox, oy, ox, oy], np.float64)
        oy, ox = int(rng.randint(0, small.shape[0] - H + 1)), int(rng.randint(0, small.shape[1] - W + 1))
        b = boxes * z - np.array([ox, oy, ox, oy], np.float64)
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        c = np.column_stack([np.clip(b[:, 0], 0, W), np.clip(b[:, 1], 0, H), np.clip(b[:, 2], 0, W), np.clip(b[:, 3], 0, H)])
        vis = (c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]) / area
        return np.ascontiguousarray(small[oy:oy + H, ox:ox + W]), c[vis > 0.7] if (vis > 0.7).any() else np.zeros((0, 4))

    def collate(self, batch):
        if not self.train:
            return torch.stack([b[0] for b in batch]), [b[1] for b in batch]
        imgs = _stack_on_device([b[0] for b in batch], self.device)
        H, W = self.size
        cm, rm = self.assigner([b[1] for b in batch], paste_boxes=[[0, 0, W, H]] * len(batch), device=self.device)
        return imgs, cm, rm
