"""WIDER FACE dataset with the reference's constructor and sample contract (tinyfaces/datasets/wider_face.py:13-239), the image
work moved to the GPU.

  * `load`                 the annotation text format of wider_face.py:65-121: name / count / count x 10 numbers, a dummy line after
                           a count of 0, |value|, zero-size boxes dropped, (x, y, w, h) -> (x1, y1, x2, y2) with the MATLAB -1.
  * `__getitem__` (train)  decodes the JPEG on the host (worker-parallel, like the reference) and returns the RAW uint8 image
                           + its boxes; everything after the decode happens in `collate` on the device: augmentation
                           (datasets/augment.py -> tf_image_prepare) and target assignment (tf_dense_overlap_targets), so a
                           batch is (img (B,3,500,500), class_map (B,25,63,63), regression_map (B,100,63,63)) as
                           wider_face.py:219-222 + the default collate produce.
  * val / test             image tensor + path, as wider_face.py:224-239.
np.random is drawn in the reference's per-sample order (scale, crop x/y, paste x/y, flip); the balance sampling inside the
target kernel uses the device RNG (documented deviation, see ops.dense_overlap_targets)."""
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import dataset

from .. import ops, transforms
from . import augment

RF = {"size": [859, 859], "stride": [8, 8], "offset": [-1, -1]}      # wider_face.py:53-55


_ATTRIBUTES = ("blur", "expression", "illumination", "invalid", "occlusion", "pose")       # columns 4..9 of a WIDER box record


def parse_annotations(path, split="train"):
    """The WIDER FACE annotation text format (what wider_face.py:65-121 reads): per image a file name, a face count n and n
    records of ten numbers `x y w h blur expression illumination invalid occlusion pose` -- or ONE all-zero record when n = 0.
    Returns one dict per image: img_path, bboxes (G, 4) float64 as (x1, y1, x2, y2) = (x, y, x + w - 1, y + h - 1) (the
    1-indexed MATLAB convention the reference keeps), and the six attribute columns; records of zero width or height are
    dropped, negative numbers lose their sign (both as the reference does).  The test split lists file names only.

    The file is tokenised ONCE; every image's records are then one reshape of a slice of a single float array."""
    with open(path) as f:
        text = f.read()
    if split == "test":
        return [{"img_path": name.strip()} for name in text.splitlines(keepends=True)]
    rows = text.split("\n")
    if rows and rows[-1] == "":
        rows.pop()
    # pass 1: where does each image start?  (a name line is followed by its count; the records of a count-0 image take one line)
    heads, at = [], 0
    while at < len(rows):
        n = int(rows[at + 1])
        heads.append((rows[at].strip(), n, at + 2))
        at += 2 + max(n, 1)
    # pass 2: every record of the file as one (R, 10) array, sliced per image
    rec_lines = [rows[first + i] for _, n, first in heads for i in range(n)]
    flat = np.abs(np.array(" ".join(rec_lines).split(), dtype=np.float64)) if rec_lines else np.empty(0)
    if flat.size != 10 * len(rec_lines):          # a record with a missing / extra field would silently shift boxes between images
        for name, n, first in heads:
            for i in range(n):
                if len(rows[first + i].split()) != 10:
                    raise ValueError(f"{path}: record {i} of {name!r} (line {first + i + 1}) has {len(rows[first + i].split())} fields, expected 10")
    flat = flat.reshape(-1, 10)
    out, cursor = [], 0
    for name, n, _ in heads:
        rec = flat[cursor:cursor + n]
        cursor += n
        rec = rec[(rec[:, 2] != 0) & (rec[:, 3] != 0)]
        corners = np.concatenate([rec[:, :2], rec[:, :2] + rec[:, 2:4] - 1.0], axis=1)
        datum = {"img_path": name, "bboxes": corners}
        datum.update({key: rec[:, 4 + j].copy() for j, key in enumerate(_ATTRIBUTES)})
        out.append(datum)
    return out


class WIDERFace(dataset.Dataset):
    def __init__(self, path, templates, img_transforms=None, dataset_root="", split="train", input_size=(500, 500),
                 heatmap_size=(63, 63), pos_thresh=0.7, neg_thresh=0.3, pos_fraction=0.5, debug=False, device="cuda", seed=0):
        super().__init__()
        self.split = split
        self.data = parse_annotations(path, split)
        print("Dataset loaded")
        print("{0} samples in the {1} dataset".format(len(self.data), self.split))
        self.templates = templates
        self.transforms = img_transforms
        self.dataset_root = Path(dataset_root)
        self.input_size, self.heatmap_size = input_size, heatmap_size
        self.pos_thresh, self.neg_thresh, self.pos_fraction = pos_thresh, neg_thresh, pos_fraction
        self.rf = RF
        self.debug = debug
        self.device = torch.device(device)
        self._step = int(seed) << 20
        self._in_stream = None

    def get_all_bboxes(self):
        return np.vstack([np.empty((0, 4))] + [d["bboxes"] for d in self.data])

    def __len__(self):
        return len(self.data)

    def _open(self, datum):
        from PIL import Image
        image_path = self.dataset_root / "WIDER_{0}".format(self.split) / "images" / datum["img_path"]
        return Image.open(image_path).convert("RGB")

    def __getitem__(self, index):
        datum = self.data[index]
        image = self._open(datum)
        if self.split == "train":
            return np.array(image, dtype=np.uint8), datum["bboxes"]
        if self.split == "val":
            return (transforms.to_tensor(image) if self.transforms is not None else image), datum["img_path"]
        return (self.transforms(image) if self.transforms is not None else image), datum["img_path"]

    def collate(self, samples):
        """Main-process half of a training batch: upload, augment on the GPU, assign targets on the GPU."""
        if self.split != "train":
            # what torch's default collate makes of [(img (3,H,W), path)] (wider_face.py:224-239 through the DataLoader of
            # datasets/__init__.py:46-50): a batch axis on the images, a list of the paths -- evaluate_model.py:60-68 reads
            # img[0] / filename[0].  Images of different sizes do not stack there either (batch_size 1 is the reference's use).
            imgs, paths = zip(*samples)
            if all(isinstance(i, torch.Tensor) for i in imgs):
                return torch.stack(list(imgs)), list(paths)
            return list(imgs), list(paths)
        mean, std = ops.IMAGE_MEAN, ops.IMAGE_STD
        ts = getattr(self.transforms, "transforms", None)
        if ts is not None and len(ts) == 2 and type(ts[1]).__name__ == "Normalize":
            mean, std = tuple(float(v) for v in ts[1].mean), tuple(float(v) for v in ts[1].std)
        B = len(samples)
        if self._in_stream is None:
            self._in_stream = torch.cuda.Stream(device=self.device)
        # the batch is built on an input stream of its own: the blocking H2D copies of images / boxes then wait for THIS
        # stream only, while the training stream is still busy with the previous step (bench.py --with-augment: 960 -> 1032 img/s)
        with torch.cuda.stream(self._in_stream):
            x = torch.empty(B, 3, *self.input_size, dtype=torch.float32, device=self.device)
            boxes, pastes, flips = [], [], []
            for i, (img, bb) in enumerate(samples):
                u8 = torch.from_numpy(img).to(self.device)      # blocking on THIS stream only: the decoded (pageable) array is not ours to keep alive
                _, b, paste, flip = augment.process_inputs(u8, bb, self.input_size, self.neg_thresh, out=x[i], mean=mean, std=std)
                boxes.append(b); pastes.append(paste); flips.append(int(flip))
            self._step += 1
            cm, rm = ops.dense_overlap_targets(boxes, self.templates, self.heatmap_size, self.rf, paste_boxes=pastes, flips=flips,
                                               seed=self._step, pos_thresh=self.pos_thresh, neg_thresh=self.neg_thresh, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self._in_stream)
        for t in (x, cm, rm):
            t.record_stream(cur)                              # the caching allocator must not recycle them under the consumer
        return x, cm, rm
