"""Call surface of the reference's tinyfaces/datasets/__init__.py (`get_dataloader`).

Dataset file I/O (WIDER images / annotation parsing, tinyfaces/datasets/wider_face.py:65-239) is
outside the hot path; what this factory adds is the device-side target assignment: batches are
(img, class_map, regression_map) exactly as WIDERFace.__getitem__ yields them
(wider_face.py:219-222), but the two maps are produced by the fused HIP dense_overlap kernel
instead of the 4-deep Python loop running in DataLoader workers."""
import numpy as np
import torch
from torch.utils import data

from .synthetic import SyntheticCrops, TargetAssigner  # noqa: F401
from .templates import load_templates


def get_dataloader(datapath, args, num_templates=25, template_file="templates.json", img_transforms=None, train=True,
                   split="train"):
    """datasets/__init__.py:11-52.  `datapath == "synthetic"` (or a path that does not exist with
    args.synthetic set) yields seeded 500x500 crops + random boxes (BASELINE.json configs[2]);
    returns (loader, templates ndarray 25x5) like the reference."""
    templates = load_templates(num_templates)
    if str(datapath) == "synthetic" or getattr(args, "synthetic", False):
        ds = SyntheticCrops(templates, length=getattr(args, "synthetic_len", 256), seed=getattr(args, "seed", 0),
                            train=train)
        loader = data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=0, collate_fn=ds.collate)
        return loader, templates
    raise NotImplementedError(
        "WIDER FACE file loading (tinyfaces/datasets/wider_face.py) is not part of the accelerated hot path in this "
        "round; use datapath='synthetic' or feed TargetAssigner with your own (image, boxes) batches.")
