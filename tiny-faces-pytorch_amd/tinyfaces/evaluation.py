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
    model = model.to(device)
    model.eval()
    nt = templates.shape[0]
    scales_list = [2 ** x for x in scales]
    image = transforms.to_pil_image(img)                                  # :40
    min_side = np.min(image.size)
    t_d = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(device)

    levels = []
    if pyramid_on_gpu:
        ms = _normalize_of(img_transforms)
        if ms is None:
            raise ValueError("pyramid_on_gpu needs img_transforms = Compose([ToTensor(), Normalize(mean, std)])")
        u8 = torch.from_numpy(np.array(image, dtype=np.uint8)).to(device)      # (H, W, 3), the only upload of the image
        w, h = image.size
        for scale in scales_list:
            size = int(min_side * scale)                                       # :46 -> transforms.resize(image, int)
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = size, int(size * long / short)
            new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
            levels.append((scale, ops.image_prepare(u8, resized_hw=(new_h, new_w), mean=ms[0], std=ms[1]).unsqueeze(0)))
    else:
        # the resize + normalise of :46-53 is host work: do it for every level first, then keep the GPU busy
        for scale in scales_list:
            scaled = transforms.resize(image, int(min_side * scale))
            x = img_transforms(scaled).unsqueeze(0).float()
            levels.append((scale, x))
    cap = sum(((x.shape[2] + 7) // 8) * ((x.shape[3] + 7) // 8) for _, x in levels) * nt
    dets = torch.empty(max(cap, 1), 5, dtype=torch.float64, device=device)
    count = torch.zeros(1, dtype=torch.int32, device=device)
    biggest = max(levels, key=lambda l: l[1].shape[2] * l[1].shape[3])[1]
    # weights are constant across the pyramid (and across images when the caller already opened a session): pack once
    with torch.no_grad(), model.constant_weights(reserve=(1, biggest.shape[2], biggest.shape[3])):
        for scale, x in levels:
            out = model(x.to(device, non_blocking=True))                  # (1, 5nt, H', W')
            _, _, H, W = out.shape
            vx, vt = ops.template_masks(templates, scale, W, mask_axis)
            ops.decode_compact(out[0], t_d, torch.from_numpy(vx).to(device), torch.from_numpy(vt).to(device),
                               prob_thresh, scale, dets, count, rf)
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


def write_results(dets, img_path, split, results_dir=None):
    """evaluation.py:90-114: WIDER submission text file."""
    results_dir = Path(results_dir or f"{split}_results")
    results_dir.mkdir(parents=True, exist_ok=True)
    filename = results_dir / img_path.replace("jpg", "txt")
    filename.parent.mkdir(parents=True, exist_ok=True)
    with open(filename, "w") as f:
        f.write(img_path.split("/")[-1] + "\n")
        f.write(str(dets.shape[0]) + "\n")
        for x in dets:
            left, top = np.round(x[0]), np.round(x[1])
            width = np.round(x[2] - x[0] + 1)
            height = np.round(x[3] - x[1] + 1)
            f.write(f"{int(left)} {int(top)} {int(width)} {int(height)} {x[4]}\n")
