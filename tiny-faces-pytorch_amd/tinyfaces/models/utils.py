"""Call surface of the reference's tinyfaces/models/utils.py: get_bboxes (:4-76), regression_refinement (:79-100), balance_sampling
(:103-139), shuffle_index (:142-163).  get_bboxes is executed by the HIP decode kernel (csrc/decode.hip) and the training path samples
inside the criterion kernel (csrc/criterion.hip); the three helpers below are the reference's HOST bookkeeping under their own names, for
callers that import them (r4: rounds 1-3 only exported get_bboxes).  They consume np.random exactly like the reference, so a seeded
script draws the same samples; `balance_sampling_keep` turns those draws into the keep flags DetectionCriterion.inject_sampling takes."""
import numpy as np
import torch

from .. import ops


def get_bboxes(score_cls, score_reg, prob_cls, templates, prob_thresh, rf, scale=1, refine=True, mask_axis="w",
               device="cuda", unrefined="reference"):
    """utils.py:4-76 with numpy in / numpy out, run on the GPU.
    score_cls / prob_cls (1,H,W,nt) f32, score_reg (1,H,W,4nt) f32 -> (N,4) f64 boxes, (N,1) f32 scores.
    `prob_cls` is only used for its shape: the kernel recomputes sigmoid(score_cls) like
    evaluation.py:62.  mask_axis='w' reproduces the reference (defect D1, utils.py:44).

    refine=False (utils.py:65-66, never used by the reference's own callers): the anchor boxes without the regression
    refinement = the same kernel on zeroed regression channels (cw * exp(0) == cw, cx + cw * 0 == cx: exact).  The reference then
    takes `bboxes[0]` of a (4, N) array, i.e. it returns only the x1 ROW, shape (N,): reproduced by default (`unrefined="reference"`);
    `unrefined="boxes"` returns the (N, 4) anchor boxes that were evidently meant."""
    if not refine:
        boxes, scores = get_bboxes(score_cls, np.zeros_like(score_reg), prob_cls, templates, prob_thresh, rf, scale, True, mask_axis, device)
        return (boxes[:, 0].copy() if unrefined == "reference" else boxes), scores
    nt = templates.shape[0]
    _, H, W, _ = score_cls.shape
    vx, vt = ops.template_masks(templates, scale, W, mask_axis)
    dev = torch.device(device)
    score = torch.cat([torch.from_numpy(np.ascontiguousarray(score_cls[0].transpose(2, 0, 1))),
                       torch.from_numpy(np.ascontiguousarray(score_reg[0].transpose(2, 0, 1)))]).float().contiguous().to(dev)
    cap = H * W * nt
    dets = torch.empty(cap, 5, dtype=torch.float64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    t_d = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(dev)
    ops.decode_compact(score, t_d, torch.from_numpy(vx).to(dev), torch.from_numpy(vt).to(dev), prob_thresh, scale, dets, count, rf)
    n = int(count.item())
    out = dets[:n].cpu().numpy()
    return out[:, :4].copy(), out[:, 4:5].astype(np.float32)



def regression_refinement(tx, ty, tw, th, cx, cy, cw, ch, indices):
    """utils.py:79-100: centre / size refinement of the selected anchors.  tx..th: regression maps (any shape indexable by `indices`, the
    tuple np.where returned), cx..ch: (N,) anchor centres and sizes.  Returns the reference's (1, N, 4) array of (x1, y1, x2, y2)."""
    pick = [np.asarray(t)[indices] for t in (tx, ty, tw, th)]
    rcx, rcy = cx + cw * pick[0], cy + ch * pick[1]
    rcw, rch = cw * np.exp(pick[2]), ch * np.exp(pick[3])
    return np.stack([rcx - rcw / 2, rcy - rch / 2, rcx + rcw / 2, rcy + rch / 2], axis=-1)[None]


def shuffle_index(n, n_out):
    """utils.py:142-163: `n_out` of the indices 0..n-1 in random order (one np.random.permutation(n) draw; nothing drawn when either is 0)."""
    n, n_out = int(n), int(n_out)
    if n == 0 or n_out == 0:
        return np.empty(0)
    if n_out > n:
        raise AssertionError("shuffle_index: n_out <= n")          # the reference asserts
    return np.random.permutation(n)[:n_out]


def balance_sampling_keep(label_cls, pos_fraction, sample_size=256):
    """The draws of balance_sampling as keep flags over the C-order RANK of the positive / negative labels (uint8, 1 = the label survives):
    what DetectionCriterion.inject_sampling / tf_criterion_fwd_bwd(pos_keep, neg_keep) take.  Positives are drawn first, like the reference."""
    lab = np.asarray(label_cls)
    n_pos, n_neg = int((lab == 1).sum()), int((lab == -1).sum())
    pos_max = sample_size * pos_fraction
    pos_keep, neg_keep = np.ones(n_pos, np.uint8), np.ones(n_neg, np.uint8)
    if n_pos > pos_max:
        pos_keep[shuffle_index(n_pos, n_pos - pos_max).astype(np.int64)] = 0       # the reference draws the positives to DROP ...
    neg_max = pos_max * (1 - pos_fraction) / pos_fraction
    if n_neg > neg_max:
        neg_keep[:] = 0
        neg_keep[shuffle_index(n_neg, neg_max).astype(np.int64)] = 1               # ... and the negatives to KEEP
    return pos_keep, neg_keep


def balance_sampling(label_cls, pos_fraction, sample_size=256):
    """utils.py:103-139: at most sample_size * pos_fraction positive and the matching number of negative labels survive, the rest become 0
    ("ignore").  Mutates and returns `label_cls` (numpy array of {-1, 0, 1}, any shape)."""
    pos_keep, neg_keep = balance_sampling_keep(label_cls, pos_fraction, sample_size)
    flat = label_cls.reshape(-1)             # logical C order (a copy for a non-contiguous array: only READ here)
    pos_at, neg_at = np.flatnonzero(flat == 1), np.flatnonzero(flat == -1)
    # index-based assignment like the reference (utils.py:126-137): edits `label_cls` itself whatever its memory layout
    label_cls[np.unravel_index(pos_at[pos_keep == 0], label_cls.shape)] = 0
    label_cls[np.unravel_index(neg_at[neg_keep == 0], label_cls.shape)] = 0
    return label_cls
