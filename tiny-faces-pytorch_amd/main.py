"""Training entry point with the reference's flags (main.py:18-36) on the MI355X hot path.

    python main.py <traindata> <valdata> [--dataset-root ...]          # reference usage (Makefile:11-12)
    python main.py synthetic synthetic --epochs 1                      # seeded 500x500 crops, no dataset needed
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py synthetic synthetic   # data parallel

Additions (existing flags keep their names and defaults): --fused / --no-fused (TrainEngine vs torch.optim.SGD through
autograd), --dtype bf16|fp32, --synthetic-len.  Everything else is main.py:39-104."""
import argparse
from pathlib import Path

import torch
from torch import optim

from tinyfaces import parallel, trainer, transforms
from tinyfaces.datasets import get_dataloader
from tinyfaces.engine import TrainEngine
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel


def arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("traindata")
    parser.add_argument("valdata")
    parser.add_argument("--dataset-root", default="")
    parser.add_argument("--dataset", default="WIDERFace")
    parser.add_argument("--lr", default=1e-4, type=float)
    parser.add_argument("--weight-decay", default=0.0005, type=float)
    parser.add_argument("--momentum", default=0.9, type=float)
    parser.add_argument("--batch_size", default=12, type=int)
    parser.add_argument("--workers", default=8, type=int)
    parser.add_argument("--start-epoch", default=0, type=int)
    parser.add_argument("--epochs", default=50, type=int)
    parser.add_argument("--save-every", default=10, type=int)
    parser.add_argument("--resume", default="", help="checkpoint path (the reference declares it store_true but uses it as a path, main.py:33,74)")
    parser.add_argument("--debug", action="store_true")
    parser.add_argument("--fused", dest="fused", action="store_true", default=True)
    parser.add_argument("--no-fused", dest="fused", action="store_false")
    parser.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    parser.add_argument("--synthetic-len", dest="synthetic_len", default=240, type=int)
    return parser.parse_args()


def main():
    args = arguments()
    parallel.init_from_env()
    num_templates = 25
    normalize = transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    img_transforms = transforms.Compose([transforms.ToTensor(), normalize])
    train_loader, _ = get_dataloader(args.traindata, args, num_templates, img_transforms=img_transforms)
    model = DetectionModel(num_objects=1, num_templates=num_templates).set_compute_dtype(args.dtype)
    loss_fn = DetectionCriterion(num_templates, seed=parallel.rank(), lazy_meters=True)
    weights_dir = Path("weights")
    if not torch.cuda.is_available():
        raise SystemExit("this build of the tiny-faces hot path runs on MI355X only (no CPU fallback)")
    device = torch.device("cuda", torch.cuda.current_device())
    start_epoch = args.start_epoch
    if args.resume:
        ckpt = torch.load(args.resume, map_location="cpu")
        model.load_state_dict(ckpt["model"])
        start_epoch = start_epoch or ckpt["epoch"]
    if args.fused:
        engine = TrainEngine(model, loss_fn, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay, device=device)
        optimizer = None
    else:
        optimizer = optim.SGD(model.learnable_parameters(args.lr), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
        scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=20, last_epoch=start_epoch - 1)
    for epoch in range(start_epoch, args.epochs):
        if args.fused:
            engine.set_lr(args.lr * (0.1 ** (epoch // 20)))                    # StepLR(step_size=20), main.py:81-83
            for idx, (img, class_map, regression_map) in enumerate(train_loader):
                engine.step(img.float().to(device, non_blocking=True), class_map.float().to(device), regression_map.float().to(device))
                if parallel.rank() == 0:
                    loss_fn.flush_meters()
                    trainer.print_state(idx, epoch, len(train_loader), loss_fn.class_average.average, loss_fn.reg_average.average)
        else:
            trainer.train(model, loss_fn, optimizer, train_loader, epoch, device=device)
            scheduler.step()
        if (epoch + 1) % args.save_every == 0 and parallel.rank() == 0:
            trainer.save_checkpoint({"epoch": epoch + 1, "batch_size": train_loader.batch_size, "model": model.state_dict(),
                                     "optimizer": optimizer.state_dict() if optimizer else {}},
                                    filename="checkpoint_{0}.pth".format(epoch + 1), save_path=weights_dir)


if __name__ == "__main__":
    main()
