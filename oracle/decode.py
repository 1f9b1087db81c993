"""Oracle: score-map -> boxes decode (numpy).  Test infrastructure only.

Restates tinyfaces/models/utils.py:4-100 (`get_bboxes`, `regression_refinement`),
including defect D1 (utils.py:44 masks axis 2 = W instead of the template axis) which is
reproduced bit-for-bit when `mask_axis="w"` (the reference behaviour, default).
`mask_axis="template"` is the evidently intended behaviour (not what the reference does).
Pinned by tests/golden/decode_*.npz generated from the reference's own source.
"""
import numpy as np


def invalid_template_ids(templates, scale):
    """utils.py:17-41."""
    all_scale = np.arange(4, 12)
    one_scale = np.arange(18, 25)
    ignored = np.setdiff1d(np.arange(25), np.concatenate((all_scale, one_scale)))
    ts = templates[:, 4]
    if scale < 1:
        inv = np.where(ts[one_scale] >= 1.0)
    else:  # scale == 1 and scale > 1 branches are identical (utils.py:33-38)
        inv = np.where(ts[one_scale] != 1.0)
    return np.concatenate((ignored, one_scale[inv]))


def get_bboxes(score_cls, score_reg, prob_cls, templates, prob_thresh, rf, scale=1,
               refine=True, mask_axis="w"):
    """score_cls/prob_cls (1,H,W,nt) f32, score_reg (1,H,W,4nt) f32 -> (N,4) f64, (N,1) f32.
    NOTE mutates prob_cls in place like the reference (utils.py:44)."""
    nt = templates.shape[0]
    inv = invalid_template_ids(templates, scale)
    if mask_axis == "w":
        prob_cls[:, :, inv] = 0.0            # utils.py:44 verbatim semantics (axis 2 == W); IndexError if W <= max(inv)
    else:
        prob_cls[:, :, :, inv] = 0.0
    indices = np.where(prob_cls > prob_thresh)   # C-order over (b, y, x, t)
    fb, fy, fx, fc = indices
    scores = score_cls[fb, fy, fx, fc]
    scores = scores.reshape((scores.shape[0], 1))
    stride, offset = rf["stride"], rf["offset"]
    cy, cx = fy * stride[0] + offset[0], fx * stride[1] + offset[1]
    cw = templates[fc, 2] - templates[fc, 0] + 1
    ch = templates[fc, 3] - templates[fc, 1] + 1
    tx = score_reg[:, :, :, 0:nt]
    ty = score_reg[:, :, :, nt:2 * nt]
    tw = score_reg[:, :, :, 2 * nt:3 * nt]
    th = score_reg[:, :, :, 3 * nt:4 * nt]
    if refine:
        dcx, dcy = cw * tx[indices], ch * ty[indices]                 # utils.py:81-82
        rcx, rcy = cx + dcx, cy + dcy
        rcw = cw * np.exp(tw[indices])                                # exp evaluated in float32 (:87)
        rch = ch * np.exp(th[indices])
        rcx, rcy, rcw, rch = (a.reshape((a.shape[0], 1)) for a in (rcx, rcy, rcw, rch))
        bboxes = np.array([rcx - rcw / 2, rcy - rch / 2, rcx + rcw / 2, rcy + rch / 2]).T
    else:
        bboxes = np.array([cx - cw / 2, cy - ch / 2, cx + cw / 2, cy + ch / 2])
    bboxes = bboxes[0]
    return bboxes * (1 / scale), scores
