"""Training entry point with the reference's flags (main.py:18-36) on the MI355X hot path.

    python main.py <traindata> <valdata> [--dataset-root ...]          # reference usage (Makefile:11-12)
    python main.py synthetic synthetic --epochs 1                      # seeded 500x500 crops, no dataset needed
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py synthetic synthetic   # data parallel

Additions (existing flags keep their names and defaults): --fused / --no-fused (TrainEngine vs torch.optim.SGD through
autograd), --dtype bf16|fp32, --synthetic-len.  The schedule is the reference's: SGD(momentum, weight decay), StepLR(20),
a checkpoint every --save-every epochs (main.py:39-104)."""
import argparse
import os
from pathlib import Path

import torch
from torch import optim

from tinyfaces import parallel, trainer, transforms
from tinyfaces.datasets import get_dataloader
from tinyfaces.engine import TrainEngine
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel

NUM_TEMPLATES = 25
LR_STEP = 20            # StepLR(step_size=20, gamma=0.1), main.py:81-83

# the reference's command line (name, argparse keyword arguments), then this repo's additions
REFERENCE_FLAGS = [
    ("traindata", {}), ("valdata", {}),
    ("--dataset-root", dict(default="")), ("--dataset", dict(default="WIDERFace")),
    ("--lr", dict(default=1e-4, type=float)), ("--weight-decay", dict(default=0.0005, type=float)), ("--momentum", dict(default=0.9, type=float)),
    ("--batch_size", dict(default=12, type=int)), ("--workers", dict(default=8, type=int)),
    ("--start-epoch", dict(default=0, type=int)), ("--epochs", dict(default=50, type=int)), ("--save-every", dict(default=10, type=int)),
    ("--resume", dict(default="", help="checkpoint path (the reference declares it store_true but uses it as a path, main.py:33,74)")),
    ("--debug", dict(action="store_true")),
]
EXTRA_FLAGS = [
    ("--fused", dict(dest="fused", action="store_true", default=True)), ("--no-fused", dict(dest="fused", action="store_false")),
    ("--dtype", dict(default="bf16", choices=["bf16", "fp32"])), ("--synthetic-len", dict(dest="synthetic_len", default=240, type=int)),
    ("--seed", dict(default=0, type=int, help="synthetic data / sampling seed (each rank derives its own from it)")),
    ("--pretrained", dict(default="", help="local torchvision-format resnet101 state_dict (.pth) for the trunk: the reference starts "
                                           "from ResNet101_Weights.IMAGENET1K_V1 (model.py:13-14), which it downloads; there is no network here")),
    ("--init", dict(default="tame", choices=["tame", "kaiming"],
                    help="from-scratch runs only (no --resume / --pretrained): `tame` = the last BN of every bottleneck at 0.1 and the head weights x 0.05 "
                         "(plain kaiming saturates the sigmoid of a 101-layer random trunk: SURVEY.md 7.1); `kaiming` = torch's defaults")),
    ("--save-path", dict(dest="save_path", default="weights", help="directory of the checkpoints (the reference writes ./weights, trainer.py:20-28)")),
    ("--ohem-thresh", dict(dest="ohem_thresh", default=0.03, type=float,
                           help="loss.py:62 zeroes every label -- positives too -- whose soft-margin loss is below this (0.03: y*logit > 3.49; defect D7): a positive "
                                "the classifier is sure of also leaves the REGRESSION loss, so the boxes of a detector trained from scratch stop improving as soon "
                                "as its scores are confident (tests/test_gpu_e2e.py).  0 switches the mining off")),
]


def tame_init_(model):
    """From-scratch initialisation that trains: bn3.weight = 0.1 in every bottleneck (the residual branches start small), head weights x 0.05."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bn3.weight"):
                p.fill_(0.1)
        for name in ("score_res3", "score_res4"):
            getattr(model, name).weight.mul_(0.05)
    return model


def arguments(argv=None):
    parser = argparse.ArgumentParser()
    for name, kw in REFERENCE_FLAGS + EXTRA_FLAGS:
        parser.add_argument(name, **kw)
    return parser.parse_args(argv)


def lr_at(base_lr, epoch):
    return base_lr * 0.1 ** (epoch // LR_STEP)


def run_fused_epoch(engine, loss_fn, loader, epoch, device, base_lr):
    engine.set_lr(lr_at(base_lr, epoch))
    total = len(loader)
    for idx, batch in enumerate(loader):
        engine.step(*(t.float().to(device, non_blocking=True) for t in batch))
        if parallel.rank() == 0:
            loss_fn.flush_meters(lag=0 if idx + 1 == total else 1)      # one step behind: the host prepares batch idx + 1 while the GPU runs step idx
            trainer.print_state(idx, epoch, total, loss_fn.class_average.average, loss_fn.reg_average.average)


def load_pretrained_trunk(model, path):
    """The reference's default `pretrained_weights=ResNet101_Weights.IMAGENET1K_V1` (model.py:13-14,20) from a local file:
    a torchvision resnet101 state_dict (keys `conv1.weight`, `layer1.0...`, `fc.*`; `layer4.*` is dropped like model.py:23)
    or a checkpoint of this package / the reference (`{"model": {...}}` with `model.`-prefixed keys)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)      # a state_dict of tensors: nothing else is unpickled
    sd = sd.get("model", sd)
    if any(k.startswith("model.") for k in sd):
        missing, unexpected = model.load_state_dict(sd, strict=False)
    else:
        sd = {k: v for k, v in sd.items() if not k.startswith("layer4.")}
        missing, unexpected = model.model.load_state_dict(sd, strict=False)
    if unexpected:
        raise SystemExit(f"--pretrained {path}: unexpected keys {list(unexpected)[:5]} ... (not a resnet101 state_dict?)")
    return missing


def main():
    args = arguments()
    if not torch.cuda.is_available():
        raise SystemExit("this build of the tiny-faces hot path runs on MI355X only (no CPU fallback)")
    parallel.init_from_env(os.environ.get("TINYFACES_DIST_BACKEND"))      # (gloo: several ranks on one device, the functional tests of a 1-GPU box)
    device = torch.device("cuda", torch.cuda.current_device())
    preprocess = transforms.Compose([transforms.ToTensor(), transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    train_loader, _ = get_dataloader(args.traindata, args, NUM_TEMPLATES, img_transforms=preprocess)
    model = DetectionModel(num_objects=1, num_templates=NUM_TEMPLATES).set_compute_dtype(args.dtype)
    loss_fn = DetectionCriterion(NUM_TEMPLATES, seed=args.seed * parallel.world_size() + parallel.rank(), lazy_meters=True)
    loss_fn.ohem_thresh = args.ohem_thresh

    first_epoch = args.start_epoch
    state = None
    if args.resume:                                              # main.py:73-79 (there `--resume` is a flag used as a path: defect D3)
        state = torch.load(args.resume, map_location="cpu")
        model.load_state_dict(state["model"])
        first_epoch = first_epoch or state["epoch"]
    elif args.pretrained:
        load_pretrained_trunk(model, args.pretrained)
    else:
        if args.init == "tame":
            tame_init_(model)
        if parallel.rank() == 0:
            print(f"WARNING: training starts from RANDOM ({args.init}) weights. The reference starts from ImageNet ResNet-101 "
                  "(model.py:13-14); its lr / schedule will not reproduce its results from scratch. Pass --pretrained <resnet101.pth>.")

    engine = optimizer = scheduler = None
    if args.fused:
        engine = TrainEngine(model, loss_fn, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay, device=device)
        if state is not None:
            engine.load_optimizer_state_dict(state.get("optimizer"))      # momentum buffers (a torch.optim.SGD state_dict)
    else:
        model = model.to(device)          # before load_state_dict: torch casts the momentum buffers to the device the parameters are on THEN
        optimizer = optim.SGD(model.learnable_parameters(args.lr), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
        if state is not None and state.get("optimizer", {}).get("param_groups"):
            optimizer.load_state_dict(state["optimizer"])                 # main.py:76
        for mult, g in zip((1.0, 0.1, 1.0, 0.0), optimizer.param_groups):
            g.setdefault("initial_lr", args.lr * mult)                    # StepLR(last_epoch >= 0) requires it (main.py:81-83)
        scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=LR_STEP, last_epoch=first_epoch - 1)
        # StepLR's constructor takes one step() from the lr it finds in the groups: a checkpoint saved at a multiple of LR_STEP epochs
        # already carries the decayed lr and would be decayed AGAIN (the reference has this quirk, main.py:76-83).  Here the fused and
        # the autograd path must train a resumed run at the SAME lr: put every group on the closed form the fused engine uses.
        # Deliberate divergence from the reference on resume (documented in INTEGRATION.md): the lr stored in the checkpoint's optimizer
        # state is REPLACED by the closed form of --lr, so a checkpoint trained with a different --lr continues at the new one -- say so.
        for mult, g in zip((1.0, 0.1, 1.0, 0.0), optimizer.param_groups):
            want = lr_at(args.lr, first_epoch) * mult
            if state is not None and abs(g["lr"] - want) > 1e-12 * max(1.0, abs(want)) and abs(g["lr"] - want * 0.1) > 1e-12 and parallel.rank() == 0:
                print(f"WARNING: resumed optimizer group has lr {g['lr']:.3g}; continuing at {want:.3g} (closed form of --lr {args.lr:g} at epoch {first_epoch})")
            g["lr"] = want
        scheduler._last_lr = [g["lr"] for g in optimizer.param_groups]       # what get_last_lr() reports must be what the groups hold

    for epoch in range(first_epoch, args.epochs):
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)                                 # reshuffle the rank shards
        elif hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        if engine is not None:
            run_fused_epoch(engine, loss_fn, train_loader, epoch, device, args.lr)
        else:
            trainer.train(model, loss_fn, optimizer, train_loader, epoch, device=device)
            scheduler.step()
        done = epoch + 1
        if done % args.save_every == 0 and parallel.rank() == 0:
            if engine is not None:
                engine.set_lr(lr_at(args.lr, done))                       # what StepLR would have left in the param groups
            snapshot = {"epoch": done, "batch_size": train_loader.batch_size, "model": model.state_dict(),          # main.py:97-102
                        "optimizer": optimizer.state_dict() if optimizer is not None else engine.optimizer_state_dict(base_lr=args.lr)}
            trainer.save_checkpoint(snapshot, filename=f"checkpoint_{done}.pth", save_path=Path(args.save_path))


if __name__ == "__main__":
    main()
