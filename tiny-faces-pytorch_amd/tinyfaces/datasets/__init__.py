"""Call surface of the reference's tinyfaces/datasets/__init__.py (`get_dataloader`).

`datapath` = a WIDER FACE annotation file -> datasets/wider_face.py (parser + JPEG decode on the host, augmentation and
target assignment on the GPU); `datapath == "synthetic"` -> seeded crops.  Either way batches are
(img, class_map, regression_map) exactly as WIDERFace.__getitem__ yields them
(wider_face.py:219-222), but the two maps are produced by the fused HIP dense_overlap kernel
instead of the 4-deep Python loop running in DataLoader workers."""
import numpy as np
import torch
from torch.utils import data

from .synthetic import SyntheticCrops, SyntheticFaces, TargetAssigner  # noqa: F401
from .templates import load_templates


def get_dataloader(datapath, args, num_templates=25, template_file="templates.json", img_transforms=None, train=True,
                   split="train"):
    """datasets/__init__.py:11-52.  `datapath == "synthetic"` (or a path that does not exist with
    args.synthetic set) yields seeded 500x500 crops + random boxes (BASELINE.json configs[2]);
    returns (loader, templates ndarray 25x5) like the reference.

    Data parallel (one process per GPU, tinyfaces/parallel.py): every rank walks ITS shard of the data -- a
    DistributedSampler for annotation files, a rank-dependent seed for the synthetic crops -- so an epoch is one pass over
    the data whatever the world size (up to the sampler's padding of the last shards).  Evaluation loaders shard by stride."""
    from .. import parallel
    templates = load_templates(num_templates)
    world, rank = (parallel.world_size(), parallel.rank()) if parallel.is_distributed() else (1, 0)
    if str(datapath) == "synthetic-faces":
        # r6: a small FIXED image list with pasted faces (datasets/synthetic.py: SyntheticFaces) -- the same images for training and evaluation,
        # so that train -> checkpoint -> pyramid evaluation -> result files -> AP can be closed without WIDER assets.  Data parallel: every global
        # batch is split into disjoint rank shards (rotating, see below); evaluation: rank r takes images r, r + world, ...
        length = getattr(args, "synthetic_len", 8)
        ds = SyntheticFaces(templates, length=length, seed=getattr(args, "seed", 0), train=train, img_transforms=img_transforms)
        idx = list(range(rank, length, world)) if world > 1 else None
        if world > 1 and train:
            # the shard of a rank ROTATES with the global batch: a fixed stride would show rank 0 the even layouts only, and the BatchNorm running
            # statistics it writes into the checkpoint (per-device statistics, like the reference's) would never have seen the odd ones -- the
            # 2-rank learn-and-detect run then evaluated at AP 0.77-1.0 from run to run.  Like a DistributedSampler that reshuffles: every rank
            # meets every image, every global batch is still split into disjoint halves of equal size.
            gb = args.batch_size * world
            idx = [k for k in range(length) if (k + k // gb) % world == rank]
        loader = data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, sampler=idx, num_workers=0, collate_fn=ds.collate)
        return loader, templates
    if str(datapath) == "synthetic" or getattr(args, "synthetic", False):
        length = getattr(args, "synthetic_len", 256)
        if train:
            ds = SyntheticCrops(templates, length=max(1, length // world), seed=getattr(args, "seed", 0) * world + rank,
                                train=train, img_transforms=img_transforms)
            loader = data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, num_workers=0, collate_fn=ds.collate)
            return loader, templates
        # evaluation: ONE image list whatever the world size; rank r takes images r, r + world, ... (no padding, no duplicates), so the
        # union of the ranks' result files is exactly what a single process writes (evaluate_model.py:56-68 sharded, "replicas only")
        ds = SyntheticCrops(templates, length=length, seed=getattr(args, "seed", 0), train=False, img_transforms=img_transforms)
        loader = data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, sampler=list(range(rank, length, world)) if world > 1 else None,
                                 num_workers=0, collate_fn=ds.collate)
        return loader, templates
    # datasets/__init__.py:40-52: the WIDER FACE annotation file + image tree
    from pathlib import Path
    from .wider_face import WIDERFace
    ds = WIDERFace(Path(datapath).expanduser(), templates, split=split, img_transforms=img_transforms,
                   dataset_root=Path(getattr(args, "dataset_root", "")).expanduser(), debug=getattr(args, "debug", False))
    # decoding runs in the workers (identity collate there); the device half of a batch -- augmentation + targets -- runs in
    # this process, where the GPU context lives
    # training: equal shards (DistributedSampler pads the last ones with repeated samples when len(ds) % world != 0 -- a handful of
    # images per epoch are then seen by two ranks); evaluation: strided shards without padding, every image exactly once
    if world > 1:
        sampler = data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True) if train else list(range(rank, len(ds), world))
    else:
        sampler = None
    inner = data.DataLoader(ds, batch_size=args.batch_size, shuffle=train and sampler is None, sampler=sampler,
                            num_workers=getattr(args, "workers", 0), collate_fn=_identity)
    return DeviceCollatingLoader(inner, ds.collate), templates


def _identity(samples):
    return samples


class DeviceCollatingLoader:
    """A DataLoader whose worker processes only decode; `collate` (device work) is applied to each batch in the consumer.
    It answers everything the reference's callers ask a `torch.utils.data.DataLoader` for: `len()`, iteration,
    `.batch_size` (main.py:99), `.dataset` (`.rf`, `.transforms`: evaluate_model.py:63-64), `.sampler`."""

    def __init__(self, inner, collate):
        self.inner, self.collate, self.dataset = inner, collate, inner.dataset
        self.batch_size, self.sampler, self.num_workers = inner.batch_size, inner.sampler, inner.num_workers

    def set_epoch(self, epoch):
        """Reshuffle the rank shards (DistributedSampler.set_epoch); a no-op in a single process."""
        if hasattr(self.sampler, "set_epoch"):
            self.sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        for samples in self.inner:
            yield self.collate(samples)
