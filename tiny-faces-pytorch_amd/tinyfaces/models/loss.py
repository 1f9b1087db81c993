"""DetectionCriterion -- call surface of the reference's tinyfaces/models/loss.py:7-97, computed by
one fused HIP pass (csrc/criterion.hip): OHEM, balance sampling, masked SoftMargin + SmoothL1 sums
and d(total)/d(output), with no device->host->device round trip (loss.py:47-57 does one per step)."""
import torch
from torch import nn

from .. import ops


class AvgMeter:
    """The running per-image loss the trainer prints (attributes `average`, `num_averaged` and methods `update` / `reset` of the
    reference's meter, loss.py:7-21; it is never reset between epochs there).  A batch contributes its SUMMED loss with weight
    `size`, so the value is sum(loss) / sum(size) -- kept as that quotient's incremental update so that a printed value equals
    the reference's digit for digit (trainer two-step golden log, tests/test_gpu_model.py)."""

    __slots__ = ("average", "num_averaged")

    def __init__(self):
        self.reset()

    def reset(self):
        self.average, self.num_averaged = 0, 0

    def update(self, loss, size):
        seen = self.num_averaged
        self.num_averaged = seen + size
        self.average = (seen * self.average + float(loss)) / self.num_averaged


class _CriterionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, owner, class_map, regression_map):
        loss2, grad, labels = ops.criterion_fwd_bwd(
            output.detach(), class_map, regression_map, owner.n_templates, owner.reg_weight, owner.ohem_thresh,
            owner.max_pos, owner.max_neg, owner._pos_keep, owner._neg_keep, owner._next_seed(), want_labels=owner.keep_labels)
        ctx.save_for_backward(grad)
        owner._loss2 = loss2
        owner.sampled_class_map = labels
        total = loss2[0] + owner.reg_weight * loss2[1]
        return total.to(output.dtype)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.to(grad.dtype), None, None, None


class DetectionCriterion(nn.Module):
    """The loss for the Tiny Faces detector (loss.py:24-97).

    `lazy_meters=True` (used by the fast trainer) defers the two `float(loss)` host syncs of
    loss.py:90-91 to the moment `.class_average.average` is read."""

    def __init__(self, n_templates=25, reg_weight=1, pos_fraction=0.5, seed=0, lazy_meters=False, keep_labels=False):
        super().__init__()
        self.n_templates = n_templates
        self.reg_weight = reg_weight
        self.pos_fraction = pos_fraction
        sample_size = 256                                                  # utils.py:103
        self.max_pos = int(sample_size * pos_fraction)                     # utils.py:111
        self.max_neg = int(self.max_pos * (1 - pos_fraction) / pos_fraction)   # utils.py:126
        self.ohem_thresh = 0.03                                            # loss.py:62
        self.class_average = AvgMeter()
        self.reg_average = AvgMeter()
        self.masked_class_loss = None
        self.masked_reg_loss = None
        self.total_loss = None
        self.keep_labels = keep_labels
        self.lazy_meters = lazy_meters
        self._seed, self._calls = int(seed), 0
        self._pos_keep = self._neg_keep = None
        self._pending = []

    def _next_seed(self):
        self._calls += 1
        return (self._seed * 0x9E3779B97F4A7C15 + self._calls) & (2**64 - 1)

    def inject_sampling(self, pos_keep, neg_keep):
        """Parity hook: keep flags indexed by C-order rank of the positive / negative labels of
        each image, (B, nt*H*W) uint8 -- replays the reference's np.random.permutation draws."""
        self._pos_keep, self._neg_keep = pos_keep, neg_keep

    def flush_meters(self, lag=0):
        """Feed the loss meters from the device values of the steps enqueued so far.  `lag` = the number of MOST RECENT steps left pending: reading a
        step's loss waits for that step, so a training loop that flushes everything after every step (lag = 0) never has the next batch in preparation
        while the GPU works; with lag = 1 the host stays one step ahead and the printed averages trail by one step."""
        n = max(0, len(self._pending) - lag)
        for loss2, size in self._pending[:n]:
            v = loss2.tolist()
            self.class_average.update(v[0], size)
            self.reg_average.update(v[1], size)
        self._pending = self._pending[n:]

    def forward(self, output, class_map, regression_map):
        if not output.is_cuda:
            raise RuntimeError("DetectionCriterion: HIP kernel only (no CPU fallback)")
        if class_map.dtype != torch.float32 or not class_map.is_contiguous():
            class_map = class_map.float().contiguous()
        total = _CriterionFunction.apply(output, self, class_map, regression_map.float())
        self.total_loss = total
        loss2 = self._loss2
        self.masked_class_loss, self.masked_reg_loss = loss2[0], loss2[1]   # already summed (loss.py:87-91 only uses .sum())
        if self.lazy_meters:
            self._pending.append((loss2, output.size(0)))
        else:
            v = loss2.tolist()                                             # one sync instead of two (loss.py:90-91)
            self.class_average.update(v[0], output.size(0))
            self.reg_average.update(v[1], output.size(0))
        return total

    def reset(self):
        self.class_average.reset()
        self.reg_average.reset()
