"""Call surface of the reference's tinyfaces/trainer.py (train, save_checkpoint, print_state).

`train` keeps the reference's step order (trainer.py:72-90).  Data-parallel training hides behind
the same signature: when torch.distributed is initialised (one process per GPU, RCCL over xGMI)
gradients are averaged across ranks before optimizer.step() (tinyfaces/parallel.py)."""
from pathlib import Path

import torch

from . import parallel


def print_state(idx, epoch, size, loss_cls, loss_reg):
    """trainer.py:9-17 (same line format)."""
    if epoch >= 0:
        message = "Epoch: [{0}][{1}/{2}]\t".format(epoch, idx, size)
    else:
        message = "Val: [{0}/{1}]\t".format(idx, size)
    print(message + '\tloss_cls: {loss_cls:.6f}' '\tloss_reg: {loss_reg:.6f}'.format(loss_cls=loss_cls, loss_reg=loss_reg))


def save_checkpoint(state, filename="checkpoint.pth", save_path="weights"):
    """trainer.py:20-26."""
    if not Path(save_path).exists():
        Path(save_path).mkdir()
    torch.save(state, str(Path(save_path, filename)))


def train(model, loss_fn, optimizer, dataloader, epoch, device):
    """trainer.py:68-90."""
    model = model.to(device)
    model.train()
    reducer = parallel.reducer_for(model)
    for idx, (img, class_map, regression_map) in enumerate(dataloader):
        x = img.float().to(device, non_blocking=True)
        class_map_var = class_map.float().to(device, non_blocking=True)
        regression_map_var = regression_map.float().to(device, non_blocking=True)
        output = model(x)
        loss = loss_fn(output, class_map_var, regression_map_var)
        optimizer.zero_grad()
        loss.backward()
        if reducer is not None:
            reducer.average_gradients()
        optimizer.step()
        if not parallel.is_distributed() or parallel.rank() == 0:
            if hasattr(loss_fn, "flush_meters"):
                loss_fn.flush_meters()
            print_state(idx, epoch, len(dataloader), loss_fn.class_average.average, loss_fn.reg_average.average)
