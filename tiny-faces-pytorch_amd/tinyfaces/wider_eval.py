"""Native WIDER FACE evaluator: average precision for the easy / medium / hard settings from the
result files `write_results` produces (tinyfaces/evaluation.py:90-114) -- the step the reference delegates to the external
MATLAB/Octave `eval_tools` (Makefile:20-21, README.md:49).

PARITY UNPINNED: eval_tools is third-party MATLAB code that is neither part of the reference repository nor runnable here, and the WIDER
ground-truth .mat files are not available offline.  This module restates the published protocol (eval_tools: wider_eval.m,
evaluation.m, image_evaluation / image_pr_info / dataset_pr_info, norm_score.m, boxoverlap.m, VOCap.m):
  * scores min-max normalised over the whole prediction set;
  * per image, detections in descending score order are matched to the ground-truth box of highest IoU (+1 pixel areas);
    IoU >= 0.5 with an IGNORED box (not in the setting's keep list) removes the detection from the precision count, with a kept
    box not yet recalled it counts as a true positive;
  * precision / recall accumulated over 1000 score thresholds, AP = VOC 2010+ area under the monotone precision envelope.
Host-side numpy; tests/test_host_logic.py checks it on hand-computed cases and against a brute-force re-derivation."""
import os

import numpy as np

THRESH_NUM = 1000


def box_overlap(gt, det):
    """IoU of every row of gt (G,4: x1,y1,x2,y2) with one det box, MATLAB pixel convention (+1 on widths / heights)."""
    x1 = np.maximum(gt[:, 0], det[0]); y1 = np.maximum(gt[:, 1], det[1])
    x2 = np.minimum(gt[:, 2], det[2]); y2 = np.minimum(gt[:, 3], det[3])
    w, h = x2 - x1 + 1, y2 - y1 + 1
    inter = w * h
    a = (gt[:, 2] - gt[:, 0] + 1) * (gt[:, 3] - gt[:, 1] + 1)
    b = (det[2] - det[0] + 1) * (det[3] - det[1] + 1)
    o = inter / (a + b - inter)
    o[(w <= 0) | (h <= 0)] = 0
    return o


def image_evaluation(pred, gt, keep_mask, iou_thresh=0.5):
    """pred (N,5: x,y,w,h,score; score-descending), gt (G,4: x,y,w,h), keep_mask (G,) bool.  Returns pred_recall (N,) = number of
    kept ground-truth boxes recalled by the first h detections, proposal (N,) = 1 (counts towards precision) or -1 (ignored)."""
    N, G = pred.shape[0], gt.shape[0]
    p = pred[:, :4].astype(np.float64).copy(); p[:, 2] += p[:, 0]; p[:, 3] += p[:, 1]
    g = gt[:, :4].astype(np.float64).copy(); g[:, 2] += g[:, 0]; g[:, 3] += g[:, 1]
    recall = np.zeros(G, dtype=np.int64)
    proposal = np.ones(N, dtype=np.int64)
    pred_recall = np.zeros(N, dtype=np.int64)
    for h in range(N):
        ov = box_overlap(g, p[h])
        idx = int(np.argmax(ov))
        if ov[idx] >= iou_thresh:
            if not keep_mask[idx]:
                recall[idx] = -1
                proposal[h] = -1
            elif recall[idx] == 0:
                recall[idx] = 1
        pred_recall[h] = int((recall == 1).sum())
    return pred_recall, proposal


def image_pr_info(scores, proposal, pred_recall, thresh_num=THRESH_NUM):
    """Per threshold t (1 - t/thresh_num, t = 1..thresh_num): (# counted detections with score >= t, # recalled boxes)."""
    out = np.zeros((thresh_num, 2), dtype=np.float64)
    counted = np.cumsum(proposal == 1)
    for t in range(thresh_num):
        thresh = 1 - (t + 1) / thresh_num
        idx = np.nonzero(scores >= thresh)[0]
        if idx.size:
            r = idx[-1]
            out[t, 0] = counted[r]
            out[t, 1] = pred_recall[r]
    return out


def voc_ap(rec, prec):
    """VOCap.m: area under the monotonically decreasing precision envelope."""
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    i = np.nonzero(mrec[1:] != mrec[:-1])[0] + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]))


def norm_scores(preds):
    """norm_score.m: min-max over every score of the prediction set.  preds: {event: {image: (N,5)}} -> new dict."""
    allv = [v[:, 4] for ev in preds.values() for v in ev.values() if v.shape[0]]
    if not allv:
        return preds
    lo, hi = min(float(v.min()) for v in allv), max(float(v.max()) for v in allv)
    d = hi - lo if hi > lo else 1.0
    return {e: {k: np.column_stack([v[:, :4], (v[:, 4] - lo) / d]) if v.shape[0] else v for k, v in ev.items()} for e, ev in preds.items()}


def evaluate_setting(preds, gt_boxes, keep_lists, iou_thresh=0.5, thresh_num=THRESH_NUM):
    """preds {event: {image: (N,5) score-descending, normalised}}, gt_boxes {event: {image: (G,4) x,y,w,h}},
    keep_lists {event: {image: 0-based indices of the boxes that count in this setting}}.  Returns (AP, pr_curve (T,2))."""
    pr = np.zeros((thresh_num, 2))
    count_face = 0
    for e, images in gt_boxes.items():
        for name, gt in images.items():
            keep = np.asarray(keep_lists.get(e, {}).get(name, []), dtype=np.int64).reshape(-1)
            count_face += keep.size
            pred = preds.get(e, {}).get(name)
            if gt.shape[0] == 0 or pred is None or pred.shape[0] == 0:
                continue
            mask = np.zeros(gt.shape[0], dtype=bool)
            mask[keep] = True
            pred_recall, proposal = image_evaluation(pred, gt, mask, iou_thresh)
            pr += image_pr_info(pred[:, 4], proposal, pred_recall, thresh_num)
    precision = np.divide(pr[:, 1], pr[:, 0], out=np.zeros(thresh_num), where=pr[:, 0] > 0)      # MATLAB's max() skips the 0/0 NaNs
    recall = pr[:, 1] / max(count_face, 1)
    return voc_ap(recall, precision), np.stack([precision, recall], 1)


def read_predictions(pred_dir):
    """The tree write_results creates: <pred_dir>/<event>/<image>.txt = name, count, then `x y w h score` rows."""
    preds = {}
    for event in sorted(os.listdir(pred_dir)):
        ed = os.path.join(pred_dir, event)
        if not os.path.isdir(ed):
            continue
        preds[event] = {}
        for fn in sorted(os.listdir(ed)):
            if not fn.endswith(".txt"):
                continue
            lines = open(os.path.join(ed, fn)).read().split("\n")
            n = int(lines[1]) if len(lines) > 1 and lines[1].strip() else 0
            rows = np.array([[float(v) for v in l.split()[:5]] for l in lines[2:2 + n] if l.strip()], dtype=np.float64).reshape(-1, 5)
            order = np.argsort(-rows[:, 4], kind="stable")
            preds[event][fn[:-4]] = rows[order]
    return preds


def load_ground_truth(gt_dir, setting):
    """wider_face_val.mat (face_bbx_list, event_list, file_list) + wider_<setting>_val.mat (gt_list, 1-based indices)."""
    from scipy.io import loadmat
    base = loadmat(os.path.join(gt_dir, "wider_face_val.mat"))
    sub = loadmat(os.path.join(gt_dir, f"wider_{setting}_val.mat"))
    gt_boxes, keep = {}, {}
    for i in range(len(base["event_list"])):
        event = str(base["event_list"][i][0][0])
        gt_boxes[event], keep[event] = {}, {}
        files, boxes, kl = base["file_list"][i][0], base["face_bbx_list"][i][0], sub["gt_list"][i][0]
        for j in range(len(files)):
            name = str(files[j][0][0])
            gt_boxes[event][name] = np.asarray(boxes[j][0], dtype=np.float64).reshape(-1, 4)
            keep[event][name] = np.asarray(kl[j][0], dtype=np.int64).reshape(-1) - 1
    return gt_boxes, keep


def evaluate(pred_dir, gt_dir, settings=("easy", "medium", "hard"), iou_thresh=0.5):
    """AP per setting for a directory of result files against the WIDER val ground truth."""
    preds = norm_scores(read_predictions(pred_dir))
    return {s: evaluate_setting(preds, *load_ground_truth(gt_dir, s), iou_thresh=iou_thresh)[0] for s in settings}


if __name__ == "__main__":      # python -m tinyfaces.wider_eval <results dir> <eval_tools/ground_truth dir>   (replaces `make evaluate`'s MATLAB step)
    import sys
    for setting, ap in evaluate(sys.argv[1], sys.argv[2]).items():
        print(f"{setting:7s} AP = {ap:.4f}")
