"""Evaluation entry point with the reference's flags (evaluate_model.py:17-31) on the MI355X hot path.
`dataset` = a WIDER val/test annotation file (+ --dataset-root), as in the reference's `make evaluate`, or `synthetic` for seeded
random images (no WIDER files needed); results are written in the WIDER submission format by write_results exactly like
evaluate_model.py:47-68.  The pyramid levels are built on the GPU."""
import argparse

import torch

from types import SimpleNamespace

from tinyfaces import ops, transforms
from tinyfaces.datasets import get_dataloader
from tinyfaces.datasets.templates import load_templates
from tinyfaces.evaluation import get_detections, get_model, write_results


# the reference's command line (evaluate_model.py:17-31) + one addition for the synthetic data set
FLAGS = [
    ("dataset", {}), ("--split", dict(default="val")), ("--dataset-root", {}),
    ("--checkpoint", dict(default="", help="The path to the model checkpoint")),
    ("--prob_thresh", dict(type=float, default=0.03)), ("--nms_thresh", dict(type=float, default=0.3)),
    ("--workers", dict(default=8, type=int)), ("--batch_size", dict(default=1, type=int)),
    ("--results_dir", dict(default=None)), ("--debug", dict(action="store_true")),
    ("--num-images", dict(default=4, type=int, help="synthetic dataset only")),
]


def arguments(argv=None):
    parser = argparse.ArgumentParser("Model Evaluator")
    for name, kw in FLAGS:
        parser.add_argument(name, **kw)
    return parser.parse_args(argv)


def main():
    args = arguments()
    if not torch.cuda.is_available():
        raise SystemExit("this build of the tiny-faces hot path runs on MI355X only (no CPU fallback)")
    device = torch.device("cuda:0")
    templates = load_templates()
    model = get_model(args.checkpoint, num_templates=templates.shape[0])
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    model = model.to(device).eval()
    if args.dataset != "synthetic":
        largs = SimpleNamespace(batch_size=1, workers=args.workers, dataset_root=args.dataset_root or "", debug=args.debug)
        loader, templates = get_dataloader(args.dataset, largs, train=False, split=args.split, img_transforms=tf)
        with torch.no_grad(), model.constant_weights():
            for img, filename in loader:                      # (3, H, W) float in [0, 1] (val) and its path, wider_face.py:224-239
                dets = get_detections(model, img, templates, loader.dataset.rf, tf, args.prob_thresh, args.nms_thresh, device=device,
                                      pyramid_on_gpu=True)
                write_results(dets, filename, args.split, args.results_dir)
        return
    g = torch.Generator().manual_seed(0)
    with torch.no_grad(), model.constant_weights():          # the checkpoint does not change between images: pack the weights once
        for i in range(args.num_images):
            img = torch.rand(3, 960, 1280, generator=g)          # wide enough for the 2^-2 level: the W-axis mask of defect D1 needs W' >= 25
            dets = get_detections(model, img, templates, ops.RF, tf, args.prob_thresh, args.nms_thresh, device=device)
            write_results(dets, f"synthetic/img_{i}.jpg", args.split, args.results_dir)
            print(f"img_{i}: {dets.shape[0]} detections")


if __name__ == "__main__":
    main()
