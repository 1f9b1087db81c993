"""FusedSGD -- torch.optim.SGD(momentum, weight_decay) semantics (main.py:67-70) executed by one HIP
kernel launch per parameter GROUP instead of one multi-tensor pass per op, when the group's
parameters and gradients are adjacent slices of flat buffers (DetectionModel.flatten_parameters()).
Falls back to per-tensor launches of the same kernel otherwise -- never to torch math."""
import torch

from . import ops


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            lr, mu, wd = group["lr"], group["momentum"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.sgd_step(p.data, g, st["momentum_buffer"], lr, mu, wd)
        return None
