"""Oracle: DetectionCriterion (OHEM + balance sampling + masked SoftMargin/SmoothL1 sum).

Restates tinyfaces/models/loss.py:29-93 and tinyfaces/models/utils.py:103-163
(`balance_sampling`, `shuffle_index`).  torch-CPU / numpy.  Test infrastructure only.
Pinned by tests/golden/criterion_*.npz generated from the reference's own source.

The `np.random.permutation` draws (utils.py:155) are consumed in the same order as the
reference: per image, first the positive drop (if n_pos > 128) then the negative keep
(if n_neg > 128); `selection_masks` exposes the resulting keep decisions in C-order rank
space so that they can be injected into the HIP kernel.
"""
import numpy as np
import torch
import torch.nn.functional as F


def shuffle_index(n, n_out, rng=np.random):
    n, n_out = int(n), int(n_out)
    if n == 0 or n_out == 0:
        return np.empty(0)
    x = rng.permutation(n)
    assert n_out <= n
    return x[:n_out] if n_out != n else x


def balance_sampling(label_cls, pos_fraction=0.5, sample_size=256, rng=np.random, record=None):
    """utils.py:103-139.  Mutates and returns label_cls (nt, H, W).
    `record` (dict) receives 'pos_keep'/'neg_keep' uint8 arrays over the C-order rank of the
    positive / negative entries (1 = label survives)."""
    pos_maxnum = sample_size * pos_fraction
    pos_idx = np.flatnonzero(label_cls == 1)
    pos_keep = np.ones(pos_idx.size, dtype=np.uint8)
    if pos_idx.size > pos_maxnum:
        didx = shuffle_index(pos_idx.size, pos_idx.size - pos_maxnum, rng).astype(np.int64)
        label_cls.reshape(-1)[pos_idx[didx]] = 0
        pos_keep[didx] = 0
    neg_maxnum = pos_maxnum * (1 - pos_fraction) / pos_fraction
    neg_idx = np.flatnonzero(label_cls == -1)
    neg_keep = np.ones(neg_idx.size, dtype=np.uint8)
    if neg_idx.size > neg_maxnum:
        ridx = shuffle_index(neg_idx.size, neg_maxnum, rng).astype(np.int64)
        didx = np.delete(np.arange(0, neg_idx.size), ridx)
        label_cls.reshape(-1)[neg_idx[didx]] = 0
        neg_keep[didx] = 0
    if record is not None:
        record["pos_keep"], record["neg_keep"] = pos_keep, neg_keep
    return label_cls


def criterion(output, class_map, regression_map, n_templates=25, reg_weight=1,
              pos_fraction=0.5, rng=np.random, want_grad=True):
    """loss.py:65-93.  output (B,5nt,H,W) f32 tensor; class_map (B,nt,H,W) f32 (cloned, the
    reference mutates it in place); regression_map (B,4nt,H,W) f32.
    Returns dict(total, cls, reg, class_map_final, grad, records)."""
    output = output.detach().clone().requires_grad_(want_grad)
    class_map = class_map.clone().contiguous()      # balance_sampling edits lab[b].reshape(-1) in place: needs a C-ordered view
    classification = output[:, 0:n_templates]
    regression = output[:, n_templates:]
    # hard negative mining (loss.py:59-63)
    l = F.soft_margin_loss(classification.detach(), class_map, reduction="none")
    class_map[l < 0.03] = 0
    # balance sampling (loss.py:47-57)
    lab = class_map.numpy()
    records = []
    for b in range(lab.shape[0]):
        rec = {}
        lab[b] = balance_sampling(lab[b], pos_fraction, rng=rng, record=rec)
        records.append(rec)
    class_map = torch.from_numpy(lab)
    class_loss = F.soft_margin_loss(classification, class_map, reduction="none")
    class_mask = (class_map != 0).type(output.dtype)
    masked_cls = class_mask * class_loss
    reg_loss = F.smooth_l1_loss(regression, regression_map, reduction="none")
    reg_mask = (class_map > 0).repeat(1, 4, 1, 1).type(output.dtype)
    masked_reg = reg_mask * reg_loss
    total = masked_cls.sum() + reg_weight * masked_reg.sum()
    grad = None
    if want_grad:
        total.backward()
        grad = output.grad.detach()
    return dict(total=float(total), cls=float(masked_cls.sum()), reg=float(masked_reg.sum()),
                class_map_final=class_map, grad=grad, records=records)
