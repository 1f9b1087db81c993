"""Oracle: image-pyramid detection loop + global NMS.  Test infrastructure only.

Restates tinyfaces/evaluation.py:20-87 (`get_detections`) on torch-CPU, with the
torchvision transforms / nms restated in oracle/refstub.py and oracle/nms.py.
Returns (K,5): the reference's (K,4) boxes (evaluation.py:85-87, defect D2 drops the
score) with the score re-attached as column 4; rows and order identical.
"""
import numpy as np
import torch

from oracle.decode import get_bboxes
from oracle.nms import nms
from oracle.refstub import resize, to_pil_image


def get_detections(model, img, templates, rf, img_transforms, prob_thresh=0.65, nms_thresh=0.3,
                   scales=(-2, -1, 0, 1), mask_axis="w", return_candidates=False):
    model.eval()
    dets = np.empty((0, 5))
    nt = templates.shape[0]
    image = to_pil_image(img)                                            # evaluation.py:40
    min_side = np.min(image.size)
    with torch.no_grad():
        for scale in [2 ** x for x in scales]:
            scaled = resize(image, int(min_side * scale))                # :46-47
            x = img_transforms(scaled).unsqueeze(0).float()
            out = model(x)
            score_cls = out[:, :nt]
            prob_cls = torch.sigmoid(score_cls)
            score_cls = score_cls.numpy().transpose((0, 2, 3, 1))
            prob_cls = prob_cls.numpy().transpose((0, 2, 3, 1)).copy()
            score_reg = out[:, nt:].numpy().transpose((0, 2, 3, 1))
            b, s = get_bboxes(score_cls, score_reg, prob_cls, templates, prob_thresh, rf, scale,
                              mask_axis=mask_axis)
            dets = np.vstack((dets, np.hstack((b, s))))                  # :76-78 (promotes to f64)
    keep = nms(dets[:, :4], dets[:, 4], nms_thresh)                      # :80-84
    if return_candidates:
        return dets[keep], dets, keep
    return dets[keep]
