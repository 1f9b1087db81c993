"""Call surface of the reference's tinyfaces/models/utils.py (get_bboxes, regression_refinement,
balance_sampling, shuffle_index).  get_bboxes is executed by the HIP decode kernel
(csrc/decode.hip); balance sampling lives inside the criterion kernel (csrc/criterion.hip)."""
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
