"""Call surface of the reference's tinyfaces/trainer.py (train, save_checkpoint, print_state).

`train` keeps the reference's step order (trainer.py:72-90).  Data-parallel training hides behind
the same signature: when torch.distributed is initialised (one process per GPU, RCCL over xGMI)
gradients are averaged across ranks before optimizer.step() (tinyfaces/parallel.py)."""
from pathlib import Path

import torch

from . import parallel

_LOSS_FMT = "\tloss_cls: {:.6f}\tloss_reg: {:.6f}"


def print_state(idx, epoch, size, loss_cls, loss_reg):
    """One progress line in the reference's format (trainer.py:9-17): training lines carry the epoch, validation lines
    (epoch < 0) do not."""
    head = f"Epoch: [{epoch}][{idx}/{size}]\t" if epoch >= 0 else f"Val: [{idx}/{size}]\t"
    print(head + _LOSS_FMT.format(loss_cls, loss_reg))


def save_checkpoint(state, filename="checkpoint.pth", save_path="weights"):
    """trainer.py:20-26: `state` goes to <save_path>/<filename>; the directory is created on first use."""
    target = Path(save_path)
    target.mkdir(exist_ok=True)
    torch.save(state, str(target / filename))


def _is_logging_rank():
    return parallel.rank() == 0 or not parallel.is_distributed()


def train(model, loss_fn, optimizer, dataloader, epoch, device):
    """One epoch with the reference's step order (trainer.py:68-90): forward, criterion, zero_grad, backward, [all-reduce],
    optimizer step, progress line."""
    net = model.to(device).train()
    reducer = parallel.reducer_for(net)
    n_batches = len(dataloader)
    for step, batch in enumerate(dataloader):
        image, cls_target, reg_target = (t.float().to(device, non_blocking=True) for t in batch)
        loss = loss_fn(net(image), cls_target, reg_target)
        optimizer.zero_grad()
        loss.backward()
        if reducer is not None:
            reducer.average_gradients()
        optimizer.step()
        if _is_logging_rank():
            flush = getattr(loss_fn, "flush_meters", None)
            if flush is not None:
                flush()
            print_state(step, epoch, n_batches, loss_fn.class_average.average, loss_fn.reg_average.average)
        elif getattr(loss_fn, "_pending", None):
            loss_fn._pending.clear()            # lazy meters are only ever read on the logging rank: do not pin device scalars elsewhere
