"""Host-side wrappers of the HIP entry points (device tensors in, device tensors out).

Each function is the MI355X replacement of the reference call cited in its docstring
(file:line relative to the reference repo).  PyTorch only supplies memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import _hip
from ._hip import check, lib, ptr, require_gpu, stream

RF = {"size": [859, 859], "stride": [8, 8], "offset": [-1, -1]}      # tinyfaces/datasets/wider_face.py:55

_ws_cache = {}


def _workspace(key, nbytes, device):
    """Grow-only scratch buffers keyed by purpose + device (kernels never allocate)."""
    k = (key, str(device))
    buf = _ws_cache.get(k)
    # grow-only, except that a buffer of more than 1 GiB is given back once a request needs less than a quarter of it (one unusually
    # long candidate list must not pin gigabytes of NMS bit matrix for the rest of the process)
    if buf is None or buf.numel() < nbytes or (buf.numel() > (1 << 30) and nbytes * 4 < buf.numel()):
        _ws_cache.pop(k, None)
        buf = None
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[k] = buf
    return buf


# --------------------------------------------------------------------------- targets
def dense_overlap_targets(boxes_per_image, templates, heatmap_size=(63, 63), rf=RF, paste_boxes=None, flips=None,
                          noise=None, seed=0, pos_thresh=0.7, neg_thresh=0.3, device=None):
    """compute_dense_overlap + DataProcessor.get_padding/get_regression/get_heatmaps fused
    (tinyfaces/datasets/dense_overlap.py:4-75, tinyfaces/datasets/processor.py:114-277).

    boxes_per_image: list of (G_i, 4) float64 arrays/tensors (x1,y1,x2,y2).
    noise: optional list of (vsy,vsx,nt,G_i_valid) float64 arrays = the np.random.rand draw of
           processor.py:195 (for bit-parity tests); default: device counter RNG seeded by `seed`.
    Returns class_map (B,nt,vsy,vsx) f32 and regression_map (B,4nt,vsy,vsx) f32 on `device`
    -- the CHW layout WIDERFace.__getitem__ hands to the trainer (wider_face.py:186-187)."""
    device = torch.device(device if device is not None else "cuda")
    if device.type != "cuda":
        raise RuntimeError("dense_overlap_targets: HIP kernel only (no CPU fallback)")
    B = len(boxes_per_image)
    vsy, vsx = heatmap_size
    t = torch.as_tensor(np.asarray(templates), dtype=torch.float64)
    nt, tstride = t.shape
    kept, offs = [], [0]
    for b in boxes_per_image:
        b = torch.as_tensor(np.asarray(b.cpu() if torch.is_tensor(b) else b), dtype=torch.float64).reshape(-1, 4)
        valid = ~((b[:, 2] <= b[:, 0]) | (b[:, 3] <= b[:, 1]))          # processor.py:228-232
        kept.append(b[valid])
        offs.append(offs[-1] + int(valid.sum()))
    total = offs[-1]
    boxes_d = (torch.cat(kept) if total else torch.zeros(1, 4, dtype=torch.float64)).contiguous().to(device)
    offs_d = torch.tensor(offs, dtype=torch.int32, device=device)
    t_d = t.contiguous().to(device)
    paste_d = None if paste_boxes is None else torch.as_tensor(np.asarray(paste_boxes), dtype=torch.int32).reshape(B, 4).to(device)
    flips_d = None if flips is None else torch.as_tensor(np.asarray(flips), dtype=torch.int32).reshape(B).to(device)
    noise_d = noff_d = None
    if noise is not None:
        flat, noff = [], [0]
        for i, nz in enumerate(noise):
            nz = np.ascontiguousarray(np.asarray(nz, dtype=np.float64))
            assert nz.shape == (vsy, vsx, nt, offs[i + 1] - offs[i]), nz.shape
            flat.append(nz.reshape(-1))
            noff.append(noff[-1] + nz.size)
        cat = np.concatenate(flat) if noff[-1] else np.zeros(1)
        noise_d = torch.from_numpy(cat).to(device)
        noff_d = torch.tensor(noff, dtype=torch.int64, device=device)
    cls = torch.empty(B, nt, vsy, vsx, dtype=torch.float32, device=device)
    reg = torch.empty(B, 4 * nt, vsy, vsx, dtype=torch.float32, device=device)
    wsb = lib().tf_targets_workspace_bytes(total)
    ws = _workspace("targets", wsb, device)
    ofy, ofx = rf["offset"]
    sty, stx = rf["stride"]
    with torch.cuda.device(device):
        check(lib().tf_dense_overlap_targets(ptr(boxes_d), ptr(offs_d), B, ptr(t_d), nt, tstride, vsy, vsx, ofy, ofx, sty, stx,
                                             ptr(paste_d), ptr(flips_d), ptr(noise_d), ptr(noff_d), int(seed) & (2**64 - 1),
                                             float(pos_thresh), float(neg_thresh), ptr(cls), ptr(reg), ptr(ws), wsb, stream()),
              "tf_dense_overlap_targets")
    return cls, reg


def dense_overlap_targets_device(boxes_d, offs_d, total_boxes, templates_d, heatmap_size=(63, 63), rf=RF, paste_d=None, flips_d=None,
                                 seed=0, pos_thresh=0.7, neg_thresh=0.3, out=None):
    """Same kernel as dense_overlap_targets with every operand already resident in HBM:
    boxes_d (total,4) f64 (degenerate boxes removed), offs_d (B+1,) i32, templates_d (nt,5) f64, paste_d (B,4) i32."""
    require_gpu(boxes_d, "dense_overlap_targets_device")
    B = offs_d.numel() - 1
    vsy, vsx = heatmap_size
    nt, tstride = templates_d.shape
    dev = boxes_d.device
    if out is None:
        out = (torch.empty(B, nt, vsy, vsx, dtype=torch.float32, device=dev), torch.empty(B, 4 * nt, vsy, vsx, dtype=torch.float32, device=dev))
    cls, reg = out
    wsb = lib().tf_targets_workspace_bytes(total_boxes)
    ws = _workspace("targets", wsb, dev)
    ofy, ofx = rf["offset"]
    sty, stx = rf["stride"]
    check(lib().tf_dense_overlap_targets(ptr(boxes_d), ptr(offs_d), B, ptr(templates_d), nt, tstride, vsy, vsx, ofy, ofx, sty, stx,
                                         ptr(paste_d), ptr(flips_d), None, None, int(seed) & (2**64 - 1), float(pos_thresh),
                                         float(neg_thresh), ptr(cls), ptr(reg), ptr(ws), wsb, stream()), "tf_dense_overlap_targets")
    return cls, reg


def dense_overlap_iou(boxes, templates, heatmap_size=(63, 63), rf=RF, device="cuda"):
    """Raw rounded IoU tensor (vsy,vsx,nt,G) f64 of compute_dense_overlap (dense_overlap.py:4-75); test hook."""
    vsy, vsx = heatmap_size
    b = torch.as_tensor(np.asarray(boxes), dtype=torch.float64).reshape(-1, 4).contiguous().to(device)
    t = torch.as_tensor(np.asarray(templates), dtype=torch.float64).contiguous().to(device)
    G = b.shape[0]
    out = torch.zeros(vsy, vsx, t.shape[0], G, dtype=torch.float64, device=device)
    if G:
        ofy, ofx = rf["offset"]
        sty, stx = rf["stride"]
        check(lib().tf_dense_overlap_iou(ptr(b), G, ptr(t), t.shape[0], t.shape[1], vsy, vsx, ofy, ofx, sty, stx, ptr(out), stream()),
              "tf_dense_overlap_iou")
    return out


def pairwise_iou_distance(boxes, device="cuda"):
    """1 - IoU for every pair of (x1, y1, x2, y2) float64 boxes: the distance matrix of the template clustering
    (tinyfaces/clustering/cluster.py:28-37), (n, n) float64 device tensor, bit-exact with the reference's double loop."""
    b = torch.as_tensor(np.asarray(boxes), dtype=torch.float64).reshape(-1, 4).contiguous().to(device)
    require_gpu(b, "pairwise_iou_distance")
    n = b.shape[0]
    out = torch.empty(n, n, dtype=torch.float64, device=b.device)
    with torch.cuda.device(b.device):
        check(lib().tf_pairwise_iou_distance(ptr(b), n, ptr(out), stream()), "tf_pairwise_iou_distance")
    return out


# --------------------------------------------------------------------------- NMS
def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics on float64 device tensors (call site tinyfaces/evaluation.py:84).
    Returns int64 indices of kept boxes in descending-score order."""
    require_gpu(boxes, "nms")
    boxes = boxes.to(torch.float64).contiguous()
    scores = scores.to(torch.float64).contiguous()
    n = boxes.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=boxes.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    wsb = lib().tf_nms_workspace_bytes(n)
    ws = _workspace("nms", wsb, boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib().tf_nms_f64(ptr(boxes), ptr(scores), n, float(iou_threshold), ptr(keep), ptr(cnt), ptr(ws), wsb, stream()), "tf_nms_f64")
    return keep[: int(cnt.item())]


NMS_MAX_SEGMENTS = 64                                     # TF_NMS_MAX_SEGMENTS (include/tinyfaces_hip.h)
NMS_MASK_BUDGET_BYTES = 2 << 30                           # suppression bit matrices of ONE tf_nms_f64_batched call (n_s^2 / 8 bytes per segment)


def _mask_bytes(n):
    return n * ((n + 63) // 64) * 8


def nms_batched(boxes, scores, seg_offsets, iou_threshold, mask_budget_bytes=None):
    """`_nms_batched_call` over groups of consecutive segments whose bit matrices fit `mask_budget_bytes` (default 2 GiB): at the
    evaluation default prob_thresh = 0.03 an image can have tens of thousands of candidates (300 MB of matrix each), so a 64-image
    batch in one call would ask for tens of GB of workspace.  A single segment larger than the budget still goes alone."""
    offs = [int(o) for o in seg_offsets]
    budget = NMS_MASK_BUDGET_BYTES if mask_budget_bytes is None else int(mask_budget_bytes)
    S = len(offs) - 1
    sizes = [_mask_bytes(b - a) for a, b in zip(offs, offs[1:])]
    if S < 1 or sum(sizes) <= budget:
        return _nms_batched_call(boxes, scores, offs, iou_threshold)
    out, first, acc = [], 0, 0
    for s in range(S + 1):
        if s == S or (s > first and acc + sizes[s] > budget):
            a, b = offs[first], offs[s]
            keeps = _nms_batched_call(boxes[a:b], scores[a:b], [o - a for o in offs[first:s + 1]], iou_threshold)
            out += [k + a for k in keeps]                     # indices into the caller's concatenated input
            first, acc = s, 0
        if s < S:
            acc += sizes[s]
    return out


def _nms_batched_call(boxes, scores, seg_offsets, iou_threshold):
    """S independent NMS problems in one call (BASELINE.json configs[4]: batched multi-scale NMS): segment s = rows
    [seg_offsets[s], seg_offsets[s+1]) of `boxes` / `scores` (float64, device) -- the multi-scale candidate list of image s of an
    evaluation batch (one evaluation.py:80-84 per image), or one list per pyramid level.  Returns a list of S int64 tensors: the kept
    indices INTO THE CONCATENATED INPUT of each segment, in descending-score order (torchvision.ops.nms semantics per segment)."""
    require_gpu(boxes, "nms_batched")
    offs = [int(o) for o in seg_offsets]
    S = len(offs) - 1
    if S < 1 or S > NMS_MAX_SEGMENTS or offs[0] != 0 or any(b < a for a, b in zip(offs, offs[1:])) or offs[-1] != boxes.shape[0]:
        raise ValueError(f"nms_batched: bad segment offsets {offs[:4]}... for {boxes.shape[0]} boxes (1..{NMS_MAX_SEGMENTS} segments)")
    boxes = boxes.to(torch.float64).contiguous()
    scores = scores.to(torch.float64).contiguous()
    n = offs[-1]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=boxes.device)
    cnt = torch.zeros(S, dtype=torch.int32, device=boxes.device)
    host = (C.c_int32 * (S + 1))(*offs)
    wsb = lib().tf_nms_batched_workspace_bytes(host, S)
    ws = _workspace("nms", wsb, boxes.device)
    with torch.cuda.device(boxes.device):
        check(lib().tf_nms_f64_batched(ptr(boxes), ptr(scores), host, S, float(iou_threshold), ptr(keep), ptr(cnt), ptr(ws), wsb, stream()),
              "tf_nms_f64_batched")
    counts = cnt.tolist()
    return [keep[offs[s]: offs[s] + counts[s]] for s in range(S)]


# --------------------------------------------------------------------------- decode
def template_masks(templates, scale, width, mask_axis="w"):
    """Validity masks reproducing tinyfaces/models/utils.py:17-44.  mask_axis='w' is the
    reference's behaviour (defect D1: the invalid TEMPLATE ids index the W axis, IndexError
    when W <= max id); 'template' is the evidently intended masking."""
    nt = templates.shape[0]
    all_scale = np.arange(4, 12)
    one_scale = np.arange(18, 25)
    ignored = np.setdiff1d(np.arange(25), np.concatenate((all_scale, one_scale)))
    ts = np.asarray(templates)[:, 4]
    inv = np.where(ts[one_scale] >= 1.0) if scale < 1 else np.where(ts[one_scale] != 1.0)
    invalid = np.concatenate((ignored, one_scale[inv]))
    vx, vt = np.ones(width, np.uint8), np.ones(nt, np.uint8)
    if mask_axis == "w":
        if invalid.max() >= width:
            raise IndexError(f"index {invalid.max()} is out of bounds for axis 2 with size {width}")   # same failure as utils.py:44
        vx[invalid] = 0
    else:
        vt[invalid[invalid < nt]] = 0
    return vx, vt


def decode_compact(score, templates_d, valid_x, valid_t, prob_thresh, scale, dets, count, rf=RF):
    """sigmoid + threshold + ordered compaction + regression refinement of one score map
    (tinyfaces/evaluation.py:61-78, tinyfaces/models/utils.py:46-100).  score (5nt,H,W) f32 device;
    appends rows (x1,y1,x2,y2,score) f64 to dets[count...] and advances the device counter."""
    require_gpu(score, "decode_compact")
    nt = templates_d.shape[0]
    _, H, W = score.shape
    wsb = lib().tf_decode_workspace_bytes(H, W, nt)
    ws = _workspace("decode", wsb, score.device)
    sty, stx = rf["stride"]
    ofy, ofx = rf["offset"]
    with torch.cuda.device(score.device):
        check(lib().tf_decode_compact(ptr(score), nt, H, W, ptr(templates_d), templates_d.shape[1], ptr(valid_x), ptr(valid_t),
                                      float(np.float32(prob_thresh)), float(scale), sty, stx, ofy, ofx, ptr(dets), ptr(count),
                                      dets.shape[0], ptr(ws), wsb, stream()), "tf_decode_compact")


# --------------------------------------------------------------------------- criterion
def criterion_fwd_bwd(output, class_map, regression_map, n_templates=25, reg_weight=1.0, ohem_thresh=0.03, max_pos=128,
                      max_neg=128, pos_keep=None, neg_keep=None, seed=0, want_labels=False):
    """DetectionCriterion.forward + backward (tinyfaces/models/loss.py:59-93).  class_map is mined in place.
    Returns (loss[2] f64 device tensor = [sum cls, sum reg], grad wrt output, labels or None)."""
    require_gpu(output, "criterion")
    assert output.dtype == torch.float32 and class_map.dtype == torch.float32 and regression_map.dtype == torch.float32
    output, regression_map = output.contiguous(), regression_map.contiguous()
    assert class_map.is_contiguous()
    B, C5, H, W = output.shape
    nt = n_templates
    assert C5 == 5 * nt and class_map.shape == (B, nt, H, W) and regression_map.shape == (B, 4 * nt, H, W)
    dev = output.device
    grad = torch.empty_like(output)
    loss = torch.empty(2, dtype=torch.float64, device=dev)
    labels = torch.empty_like(class_map) if want_labels else None
    wsb = lib().tf_criterion_workspace_bytes(B, nt, H, W)
    ws = _workspace("criterion", wsb, dev)
    pk = None if pos_keep is None else pos_keep.to(device=dev, dtype=torch.uint8).contiguous()
    nk = None if neg_keep is None else neg_keep.to(device=dev, dtype=torch.uint8).contiguous()
    with torch.cuda.device(dev):
        check(lib().tf_criterion_fwd_bwd(ptr(output), ptr(class_map), ptr(regression_map), B, nt, H, W, float(ohem_thresh), int(max_pos),
                                         int(max_neg), float(reg_weight), ptr(pk), ptr(nk), int(seed) & (2**64 - 1), ptr(labels),
                                         ptr(grad), ptr(loss), None, ptr(ws), wsb, stream()), "tf_criterion_fwd_bwd")
    return loss, grad, labels


# --------------------------------------------------------------------------- SGD
NMS_MAX_BOXES = 524160                                   # csrc/nms.hip: (n/64 + 2) 8-byte words must fit 64 KiB of LDS
IMAGE_MEAN, IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)        # main.py:44-46


def image_prepare(img_u8, resized_hw=None, crop=None, paste=(0, 0), flip=False, out_hw=None, mean=IMAGE_MEAN, std=IMAGE_STD,
                  out=None):
    """uint8 (H, W, 3) device image -> normalised float32 (3, OH, OW) device tensor in one HIP pass (tf_image_prepare):
    PIL-exact BILINEAR resize to `resized_hw`, window `crop` = (y, x, h, w) of the resized image pasted at `paste` = (y, x)
    on the mean colour, optional mirror, ToTensor + Normalize.  Defaults: no resize, whole image, out_hw = window size
    (the evaluation pyramid level of tinyfaces/evaluation.py:46-53); training passes the 500x500 crop parameters of
    tinyfaces/datasets/processor.py:41-76."""
    require_gpu(img_u8, "image_prepare")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3:
        raise ValueError("image_prepare: expected a uint8 (H, W, 3) tensor")
    img_u8 = img_u8.contiguous()
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    RH, RW = (H, W) if resized_hw is None else (int(resized_hw[0]), int(resized_hw[1]))
    cy, cx, ch, cw = (0, 0, RH, RW) if crop is None else [int(v) for v in crop]
    OH, OW = (ch, cw) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    if out is None:
        out = torch.empty(3, OH, OW, dtype=torch.float32, device=img_u8.device)
    a = _hip.ImagePrepareArgs()
    a.img, a.H, a.W, a.RH, a.RW = ptr(img_u8), H, W, RH, RW
    a.crop_y, a.crop_x, a.crop_h, a.crop_w = cy, cx, ch, cw
    a.paste_y, a.paste_x, a.flip, a.OH, a.OW = int(paste[0]), int(paste[1]), int(bool(flip)), OH, OW
    for c in range(3):
        a.mean[c], a.std[c] = float(mean[c]), float(std[c])
        a.bg[c] = int(np.int8(np.float64(mean[c]) * 255)) & 0xFF          # processor.py:66-71: (mean * 255).astype(int8)
    a.out = ptr(out)
    with torch.cuda.device(img_u8.device):
        check(lib().tf_image_prepare(C.byref(a), stream()), "tf_image_prepare")
    return out


def sgd_step(param, grad, momentum_buf, lr, momentum, weight_decay, grad_scale=1.0):
    """torch.optim.SGD.step for one flat fp32 segment (main.py:67-70)."""
    require_gpu(param, "sgd_step")
    assert param.dtype == grad.dtype == momentum_buf.dtype == torch.float32
    assert param.is_contiguous() and grad.is_contiguous() and momentum_buf.is_contiguous()
    with torch.cuda.device(param.device):
        check(lib().tf_sgd_step(ptr(param), ptr(grad), ptr(momentum_buf), param.numel(), float(lr), float(momentum),
                                float(weight_decay), float(grad_scale), stream()), "tf_sgd_step")


# --------------------------------------------------------------------------- conv engine (used directly by the parity tests)
def pack_weight(w_oihw, dtype, transpose=False, cols_pad=None):
    require_gpu(w_oihw, "pack_weight")
    w = w_oihw.float().contiguous()
    cout, cin, kh, kw = w.shape
    rows = cin if transpose else cout
    cols = cout if transpose else cin
    rows_pad = (rows + 127) // 128 * 128
    cols_pad = cols_pad or cols
    tfd = _hip.tf_dtype(dtype)
    out = torch.empty(rows_pad, kh * kw, cols_pad, dtype=_hip.torch_dtype(tfd), device=w.device)
    check(lib().tf_pack_weight(ptr(w), cout, cin, kh, kw, int(transpose), tfd, ptr(out), rows_pad, cols_pad, stream()), "tf_pack_weight")
    return out


def stem_conv(x_nchw, w_oihw, dtype, epi=0, scale=None, shift=None):
    """conv1 of the trunk (7x7 / stride 2 / pad 3, 3 -> 64; model.py:90) straight from the NCHW fp32 image (tf_stem_conv, r4): returns
    y (N, OH, OW, 64) of `dtype` (bf16 | fp16) [, statistic rows (rows, 2, 64) with epi = EPI_STATS].  `w_oihw` (64, 3, 7, 7) fp32."""
    require_gpu(x_nchw, "stem_conv")
    x = x_nchw.float().contiguous()
    N, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wp = pack_weight(w_oihw.reshape(64, 147, 1, 1), dtype, cols_pad=192)
    tfd = _hip.tf_dtype(dtype)
    y = torch.empty(N, OH, OW, 64, dtype=dtype, device=x.device)
    st = torch.zeros(lib().tf_get_stat_rows(), 2, 64, device=x.device) if epi == _hip.EPI_STATS else None
    rows = C.c_int(0)
    check(lib().tf_stem_conv(tfd, ptr(x), N, H, W, ptr(wp), 192, ptr(y), int(epi), ptr(scale) if scale is not None else None,
                             ptr(shift) if shift is not None else None, ptr(st) if st is not None else None, C.byref(rows), stream()), "tf_stem_conv")
    return (y, st[:rows.value]) if st is not None else y


def stem_wgrad(x_nchw, g_nhwc, x_conv=None, cA=None, cB=None, cD=None):
    """Weight gradient of conv1 straight from the image (tf_stem_wgrad, r4): x (N,3,H,W) fp32, g (N,OH,OW,64) bf16 | fp16 -> (64,3,7,7) fp32.
    With x_conv (N,OH,OW,64) and cA / cB / cD (64,) fp32 the gradient operand is cA * g + cB * x_conv + cD (the stem's BN-backward apply)."""
    require_gpu(x_nchw, "stem_wgrad")
    x = x_nchw.float().contiguous()
    N, _, H, W = x.shape
    g = g_nhwc.contiguous()
    dw = torch.zeros(64, 3, 7, 7, dtype=torch.float32, device=x.device)
    opt = [ptr(t.contiguous()) if t is not None else None for t in (x_conv, cA, cB, cD)]
    check(lib().tf_stem_wgrad(_hip.tf_dtype(g.dtype), ptr(x), N, H, W, ptr(g), *opt, ptr(dw), stream()), "tf_stem_wgrad")
    return dw


def conv2d_nhwc(x, w_packed, Cout, KH, KW, stride, pad, mode=0, out_hw=None, ldy=None, pro=None, epi=0, epi_scale=None,
                epi_shift=None, aux=None, aux2=None, aux3=None, mask=None, want_stats=False, tile=0, out=None, stats_into=None):
    """x (N,H,W,Cin) dtype bf16|f32 contiguous; returns y (N,OH,OW,ldy) [, stat partials (mtiles,2,ldy)].
    out: write into this (N,OH,OW,ldy) tensor (with aux=out and TF_EPI_RES the scattered stride-2 data gradient accumulates in place);
    stats_into: add the statistic sums into these existing rows instead of fresh zeros."""
    require_gpu(x, "conv2d_nhwc")
    N, H, W, Cin = x.shape
    if out_hw is None:
        assert mode == 0
        out_hw = ((H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1)
    OH, OW = out_hw
    ldy = ldy or (Cout + 3) // 4 * 4
    a = _hip.ConvArgs()
    a.dtype, a.mode = _hip.tf_dtype(x.dtype), mode
    a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad
    a.ldy, a.epi, a.tile = ldy, epi, tile
    y = torch.empty(N, OH, OW, ldy, dtype=x.dtype, device=x.device) if out is None else out
    a.x, a.w, a.y = ptr(x), ptr(w_packed), ptr(y)
    keep = [x, w_packed, y]
    if pro is not None:
        ps, ph, relu = pro
        a.pro_scale, a.pro_shift, a.pro_relu = ptr(ps), ptr(ph), int(relu)
        keep += [ps, ph]
    if epi_scale is not None:
        a.epi_scale, a.epi_shift = ptr(epi_scale), ptr(epi_shift)
    for name, t in (("aux", aux), ("aux2", aux2), ("aux3", aux3)):
        if t is not None:
            setattr(a, name, ptr(t))
    if mask is not None:
        a.mask_scale, a.mask_shift = ptr(mask[0]), ptr(mask[1])
    stats = None
    if want_stats:
        mt = lib().tf_conv_mtiles(C.byref(a))
        check(min(mt, 0), "tf_conv_mtiles")              # negative: the requested tile code does not take this launch
        stats = torch.zeros(mt, 2, ldy, dtype=torch.float32, device=x.device) if stats_into is None else stats_into
        a.stat_out = ptr(stats)
    with torch.cuda.device(x.device):
        check(lib().tf_conv2d(C.byref(a), stream()), "tf_conv2d")
    return (y, stats) if want_stats else y


def conv2d_wgrad(x, dy, Cin, Cout, KH, KW, stride, pad, pro=None, splitk=0, tile=0, out=None, packed=False, two_phase=False):
    """x (N,H,W,ldx), dy (N,OH,OW,lddy) -> dW (Cout,Cin,KH,KW) fp32.  two_phase: hand the all-taps 3x3 kernel a partial-tile
    workspace so that its split-K slices are summed by a second kernel instead of fp32 atomics."""
    require_gpu(x, "conv2d_wgrad")
    N, H, W, ldx = x.shape
    _, OH, OW, lddy = dy.shape
    dw = torch.zeros(Cout, Cin, KH, KW, dtype=torch.float32, device=x.device) if out is None else out
    a = _hip.WgradArgs()
    a.tile = tile
    a.packed = int(packed)
    a.dtype = _hip.tf_dtype(x.dtype)
    a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad
    a.ldx, a.lddy, a.x, a.dy, a.dw_oihw, a.dw_ld, a.splitk = ldx, lddy, ptr(x), ptr(dy), ptr(dw), Cin * KH * KW, splitk
    if pro is not None:
        a.pro_scale, a.pro_shift, a.pro_relu = ptr(pro[0]), ptr(pro[1]), int(pro[2])
    ws = None
    if two_phase:
        nbytes = lib().tf_wgrad_workspace_bytes(C.byref(a))
        if nbytes:
            ws = _workspace("wgrad3", nbytes, x.device)
            a.partial_ws, a.partial_ws_bytes = ptr(ws), nbytes
    with torch.cuda.device(x.device):
        check(lib().tf_conv2d_wgrad(C.byref(a), stream()), "tf_conv2d_wgrad")
    return dw


def conv2d_wgrad_group(problems, K, pad):
    """The weight gradients of a GROUP of convolutions in one launch (tf_conv2d_wgrad_group: every output tile reduced over all pixels
    in-block, dW overwritten).  problems: [(x (N,H,W,ldx), dy (N,H,W,lddy), Cin, Cout)] bf16, stride 1; K = 1 (pad 0; any channel counts,
    one pixel count) or K = 3 (pad 1; identical shapes).  Returns the list of dW (Cout,Cin,K,K) fp32, allocated with NaN so that an
    element the launch does not write is seen."""
    args = (_hip.WgradArgs * len(problems))()
    outs = []
    for a, (x, dy, Cin, Cout) in zip(args, problems):
        require_gpu(x, "conv2d_wgrad_group")
        N, H, W, ldx = x.shape
        dw = torch.full((Cout, Cin, K, K), float("nan"), dtype=torch.float32, device=x.device)
        a.dtype = _hip.tf_dtype(x.dtype)
        a.N, a.H, a.W, a.Cin, a.OH, a.OW, a.Cout, a.KH, a.KW, a.stride, a.pad = N, H, W, Cin, H, W, Cout, K, K, 1, pad
        a.ldx, a.lddy, a.x, a.dy, a.dw_oihw, a.dw_ld = ldx, dy.shape[3], ptr(x), ptr(dy), ptr(dw), Cin * K * K
        outs.append(dw)
    with torch.cuda.device(problems[0][0].device):
        check(lib().tf_conv2d_wgrad_group(args, len(problems), stream()), "tf_conv2d_wgrad_group")
    return outs

