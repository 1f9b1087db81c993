"""Evaluation entry point with the reference's flags (evaluate_model.py:17-31) on the MI355X hot path.
`dataset` = a WIDER val/test annotation file (+ --dataset-root), as in the reference's `make evaluate`, or `synthetic` for seeded
random images (no WIDER files needed); results are written in the WIDER submission format by write_results exactly like
evaluate_model.py:47-68.  The pyramid levels are built on the GPU."""
import argparse
import os

import torch

from types import SimpleNamespace

from tinyfaces import parallel, transforms
from tinyfaces.datasets import get_dataloader
from tinyfaces.evaluation import get_detections, get_model, write_results


# the reference's command line (evaluate_model.py:17-31) + one addition for the synthetic data set
FLAGS = [
    ("dataset", {}), ("--split", dict(default="val")), ("--dataset-root", {}),
    ("--checkpoint", dict(default="", help="The path to the model checkpoint")),
    ("--prob_thresh", dict(type=float, default=0.03)), ("--nms_thresh", dict(type=float, default=0.3)),
    ("--workers", dict(default=8, type=int)), ("--batch_size", dict(default=1, type=int)),
    ("--results_dir", dict(default=None)), ("--debug", dict(action="store_true")),
    ("--num-images", dict(default=4, type=int, help="synthetic dataset only")),
    ("--mask-axis", dict(dest="mask_axis", default="w", choices=["w", "template"],
                         help="`w` (default) = the reference's template mask as written (it indexes the heat-map WIDTH: defect D1, tinyfaces/models/utils.py:44); "
                              "`template` = the mask on the template axis it was meant for")),
]


def arguments(argv=None):
    parser = argparse.ArgumentParser("Model Evaluator")
    for name, kw in FLAGS:
        parser.add_argument(name, **kw)
    return parser.parse_args(argv)


def dataloader(args):
    """evaluate_model.py:34-45."""
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    largs = SimpleNamespace(batch_size=args.batch_size, workers=args.workers, dataset_root=args.dataset_root or "", debug=args.debug,
                            synthetic_len=args.num_images)
    return get_dataloader(args.dataset, largs, train=False, split=args.split, img_transforms=tf)


def run(model, val_loader, templates, prob_thresh, nms_thresh, device, split, results_dir=None, debug=False, mask_axis="w"):
    """evaluate_model.py:48-68, statement for statement: `img[0]` / `filename[0]` undo the batch axis the loader adds.  The
    only addition is the keyword that builds the pyramid levels on the GPU (same detections bit for bit)."""
    dets = None
    for _, (img, filename) in enumerate(val_loader):
        dets = get_detections(model, img[0], templates, val_loader.dataset.rf, val_loader.dataset.transforms, prob_thresh, nms_thresh,
                              device=device, pyramid_on_gpu=True, mask_axis=mask_axis)
        write_results(dets, filename[0], split, results_dir)
        if debug:
            print(f"{filename[0]}: {dets.shape[0]} detections")
    return dets


def main():
    args = arguments()
    if not torch.cuda.is_available():
        raise SystemExit("this build of the tiny-faces hot path runs on MI355X only (no CPU fallback)")
    # Multi-GPU evaluation = replicas only: under torchrun every rank takes a strided shard of the image list
    # (datasets.get_dataloader) and writes the result files of ITS images into the shared results tree; no collective on the
    # data path, one barrier at the end.  A plain `python evaluate_model.py` is rank 0 of a world of 1, exactly the reference's loop
    # (evaluate_model.py:56-68).  TINYFACES_DIST_BACKEND=gloo + TINYFACES_SHARE_GPU=1: several ranks on one device (functional tests).
    distributed = parallel.init_from_env(os.environ.get("TINYFACES_DIST_BACKEND"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("TINYFACES_SHARE_GPU") == "1"
    device = torch.device(f"cuda:{local % torch.cuda.device_count() if share else local}")
    torch.cuda.set_device(device)
    val_loader, templates = dataloader(args)
    model = get_model(args.checkpoint, num_templates=templates.shape[0])
    model = model.to(device).eval()
    with torch.no_grad(), model.constant_weights():          # the checkpoint does not change between images: pack the weights once
        run(model, val_loader, templates, args.prob_thresh, args.nms_thresh, device, args.split, results_dir=args.results_dir,
            debug=args.debug or args.dataset == "synthetic", mask_axis=args.mask_axis)
    if distributed:
        torch.distributed.barrier()
        if parallel.rank() == 0:
            print(f"evaluated on {parallel.world_size()} ranks")
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
