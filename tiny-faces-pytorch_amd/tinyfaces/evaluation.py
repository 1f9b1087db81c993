"""Call surface of the reference's tinyfaces/evaluation.py (get_model, get_detections,
write_results).  Per pyramid level: HIP forward -> HIP sigmoid/threshold/ordered-compaction/
refinement straight into one device candidate list; then ONE float64 HIP NMS over all levels
(evaluation.py:80-84).  Only the K surviving rows cross PCIe (the reference copies three full
score maps per level, evaluation.py:64-68)."""
from pathlib import Path

import numpy as np
import torch

from . import ops, transforms
from .models.model import DetectionModel


def get_model(checkpoint=None, num_templates=25):
    """evaluation.py:12-17."""
    model = DetectionModel(num_templates=num_templates)
    if checkpoint:
        checkpoint = torch.load(checkpoint, map_location="cpu")
        model.load_state_dict(checkpoint["model"])
    return model


def _normalize_of(img_transforms):
    """(mean, std) when img_transforms is the Compose([ToTensor, Normalize]) of evaluate_model.py:35-37, else None."""
    ts = getattr(img_transforms, "transforms", None)
    if ts is not None and len(ts) == 2 and type(ts[0]).__name__ == "ToTensor" and type(ts[1]).__name__ == "Normalize":
        return tuple(float(v) for v in ts[1].mean), tuple(float(v) for v in ts[1].std)
    return None


def _on_device(model, device):
    """model.to(device) only when something has to move: nn.Module.to walks `_apply` even when it is a no-op, and DetectionModel._apply must then
    assume that every storage moved -- the pointer tables and the packed evaluation weights of an open constant_weights() session were rebuilt on
    EVERY get_detections call (r6: 80 of the 87 ms per image of the end-to-end path)."""
    p = next(model.parameters(), None)
    if p is not None and p.device.type == device.type and (device.index is None or p.device.index == device.index):
        return model
    return model.to(device)


def _pyramid_levels(img, scales, img_transforms, pyramid_on_gpu, device):
    """evaluation.py:37-53: [(scale, normalised (1,3,H,W) tensor)] for every pyramid level of one image."""
    scales_list = [2 ** x for x in scales]
    levels = []
    if pyramid_on_gpu:
        ms = _normalize_of(img_transforms)
        if ms is None:
            raise ValueError("pyramid_on_gpu needs img_transforms = Compose([ToTensor(), Normalize(mean, std)])")
        planes = transforms.to_uint8_chw(img)                                  # the pixels of to_pil_image(img) (:40), no PIL object in between
        # the only upload of the image; the CHW -> HWC interleave is a device copy (6 ms of byte-wise numpy transposition on the host otherwise)
        u8 = torch.from_numpy(planes).to(device).permute(1, 2, 0).contiguous()
        h, w = planes.shape[1:]
        min_side = min(w, h)
        for scale in scales_list:
            size = int(min_side * scale)                                       # :46 -> transforms.resize(image, int)
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long / short)
            new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
            levels.append((scale, ops.image_prepare(u8, resized_hw=(new_h, new_w), mean=ms[0], std=ms[1]).unsqueeze(0)))
    else:
        image = transforms.to_pil_image(img)                              # :40
        min_side = np.min(image.size)
        # the resize + normalise of :46-53 is host work: do it for every level first, then keep the GPU busy
        for scale in scales_list:
            scaled = transforms.resize(image, int(min_side * scale))
            x = img_transforms(scaled).unsqueeze(0).float()
            levels.append((scale, x))
    return levels


def _level_capacity(levels, nt):
    return sum(((x.shape[2] + 7) // 8) * ((x.shape[3] + 7) // 8) for _, x in levels) * nt


def _decode_levels(model, levels, templates, t_d, rf, prob_thresh, mask_axis, dets, count, device):
    """Forward + sigmoid / threshold / ordered compaction / refinement of every level, appended to dets[count...].
    The forwards of the levels run side by side on the model's lanes (DetectionModel.forward_levels); the decodes follow in level
    order on the caller's stream, so the candidate list is the sequential loop's row for row."""
    xs = [x.to(device, non_blocking=True) for _, x in levels]
    # any callable nn.Module is accepted, like the reference's get_detections (evaluation.py:56-60): only DetectionModel has lanes
    fl = getattr(model, "forward_levels", None)
    outs = fl(xs) if fl is not None else [model(x) for x in xs]
    for (scale, x), out in zip(levels, outs):                         # (1, 5nt, H', W')
        _, _, H, W = out.shape
        vx, vt = ops.template_masks(templates, scale, W, mask_axis)
        ops.decode_compact(out[0], t_d, torch.from_numpy(vx).to(device), torch.from_numpy(vt).to(device),
                           prob_thresh, scale, dets, count, rf)


def get_detections(model, img, templates, rf, img_transforms, prob_thresh=0.65, nms_thresh=0.3, scales=(-2, -1, 0, 1),
                   device=None, mask_axis="w", return_candidates=False, pyramid_on_gpu=False):
    """evaluation.py:20-87.  Returns (K,5) float64: the reference's (K,4) rows in the same order
    with the score re-attached as column 4 (defect D2: the reference drops it although
    write_results reads x[4], evaluation.py:111).  mask_axis='w' reproduces defect D1
    (tinyfaces/models/utils.py:44); 'template' masks the template axis instead.
    pyramid_on_gpu=True: the uint8 image goes to the device once and every pyramid level is produced there
    by tf_image_prepare (Pillow-exact BILINEAR resize + ToTensor + Normalize in one pass) instead of PIL + torch on the host;
    needs img_transforms = Compose([ToTensor(), Normalize(mean, std)]); same detections bit for bit."""
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda":
        raise RuntimeError("get_detections: the detector only runs on MI355X (no CPU fallback)")
    model = _on_device(model, device)
    model.eval()
    nt = templates.shape[0]
    t_d = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(device)
    levels = _pyramid_levels(img, scales, img_transforms, pyramid_on_gpu, device)
    dets = torch.empty(max(_level_capacity(levels, nt), 1), 5, dtype=torch.float64, device=device)
    count = torch.zeros(1, dtype=torch.int32, device=device)
    biggest = max(levels, key=lambda l: l[1].shape[2] * l[1].shape[3])[1]
    # weights are constant across the pyramid (and across images when the caller already opened a session): pack once
    with torch.no_grad(), model.constant_weights(reserve=(1, biggest.shape[2], biggest.shape[3])):
        _decode_levels(model, levels, templates, t_d, rf, prob_thresh, mask_axis, dets, count, device)
    n = int(count.item())
    assert n <= dets.shape[0]
    if n > ops.NMS_MAX_BOXES:
        raise RuntimeError(f"get_detections: {n} candidates above prob_thresh={prob_thresh} exceed the {ops.NMS_MAX_BOXES} boxes one NMS call "
                           "takes (64 KiB of LDS for the suppression bitmap); a trained detector keeps a few thousand -- untrained weights?")
    cand = dets[:n]
    keep = ops.nms(cand[:, :4].contiguous(), cand[:, 4].contiguous(), nms_thresh)      # :80-84
    result = cand[keep].cpu().numpy()
    if return_candidates:
        return result, cand.cpu().numpy(), keep.cpu().numpy()
    return result


def get_detections_batch(model, imgs, templates, rf, img_transforms, prob_thresh=0.65, nms_thresh=0.3, scales=(-2, -1, 0, 1),
                         device=None, mask_axis="w", pyramid_on_gpu=False):
    """`[get_detections(model, img, ...) for img in imgs]` (evaluation.py:20-87 once per image, as evaluate_model.py:60-68 does) with
    the per-image NMS calls batched: every image's multi-scale candidates go into ONE device list, the per-image segments are
    suppressed by ONE tf_nms_f64_batched call (three launches whatever the number of images; BASELINE.json configs[4]) and the
    only host synchronisation of the whole batch is the read of the candidate counts.  Row for row identical to the loop."""
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda":
        raise RuntimeError("get_detections_batch: the detector only runs on MI355X (no CPU fallback)")
    imgs = list(imgs)
    if not 1 <= len(imgs) <= ops.NMS_MAX_SEGMENTS:
        raise ValueError(f"get_detections_batch: 1..{ops.NMS_MAX_SEGMENTS} images per call, got {len(imgs)}")
    model = _on_device(model, device)
    model.eval()
    nt = templates.shape[0]
    t_d = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(device)
    per_image = [_pyramid_levels(img, scales, img_transforms, pyramid_on_gpu, device) for img in imgs]
    dets = torch.empty(max(sum(_level_capacity(lv, nt) for lv in per_image), 1), 5, dtype=torch.float64, device=device)
    count = torch.zeros(1, dtype=torch.int32, device=device)
    marks = torch.zeros(len(imgs) + 1, dtype=torch.int32, device=device)
    biggest = max((l[1] for lv in per_image for l in lv), key=lambda x: x.shape[2] * x.shape[3])
    with torch.no_grad(), model.constant_weights(reserve=(1, biggest.shape[2], biggest.shape[3])):
        for i, levels in enumerate(per_image):
            _decode_levels(model, levels, templates, t_d, rf, prob_thresh, mask_axis, dets, count, device)
            marks[i + 1:i + 2].copy_(count, non_blocking=True)        # segment boundary, recorded on the stream
    offs = marks.tolist()                                             # the one sync
    if max(b - a for a, b in zip(offs, offs[1:])) > ops.NMS_MAX_BOXES:
        raise RuntimeError(f"get_detections_batch: an image has more than {ops.NMS_MAX_BOXES} candidates above prob_thresh={prob_thresh}")
    cand = dets[:offs[-1]]
    keeps = ops.nms_batched(cand[:, :4].contiguous(), cand[:, 4].contiguous(), offs, nms_thresh)
    return [cand[k].cpu().numpy() for k in keeps]


def write_results(dets, img_path, split, results_dir=None):
    """evaluation.py:90-114: WIDER submission text file."""
    target = Path(results_dir or f"{split}_results") / img_path.replace("jpg", "txt")
    target.parent.mkdir(parents=True, exist_ok=True)
    dets = np.asarray(dets)
    # the submission format of the WIDER evaluation tool: image name, box count, then `left top width height score` per box with
    # rounded integer geometry (width = x2 - x1 + 1: the inclusive-pixel convention) and the raw score
    geometry = np.round(np.stack([dets[:, 0], dets[:, 1], dets[:, 2] - dets[:, 0] + 1, dets[:, 3] - dets[:, 1] + 1], axis=1)) if len(dets) else ()
    lines = [img_path.rsplit("/", 1)[-1], str(len(dets))]
    lines += [" ".join(str(int(v)) for v in g) + f" {d[4]}" for g, d in zip(geometry, dets)]
    target.write_text("\n".join(lines) + "\n")
