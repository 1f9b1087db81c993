"""-m gpu: BASELINE.json configs[4], the "hard setting" of evaluation: fp16 MFMA convolutions, a pyramid whose top level has a
5000-px long side, batched multi-scale NMS with up to 65 536 candidates per list.
Reference path: tinyfaces/evaluation.py:37-47 (scale list, one forward per level at any size), :80-84 (one NMS per image)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import err, from_nhwc, q, report, to_nhwc

pytestmark = pytest.mark.gpu
H = torch.float16


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("tile", [0, 11, 12, 13, 23, 32])
@pytest.mark.parametrize("case", [(2, 17, 19, 64, 64, 1, 1, 0), (2, 20, 21, 64, 64, 3, 1, 1), (1, 21, 23, 128, 128, 3, 2, 1),
                                  (2, 15, 18, 256, 512, 1, 2, 0), (3, 63, 63, 256, 64, 1, 1, 0)])
def test_conv_fp16_forward(case, tile):
    """v_mfma_f32_16x16x32_f16 instantiation of the LDS-DMA conv: same operands rounded to fp16, fp32 accumulation."""
    from tinyfaces import ops
    N, Hh, W, Cin, Cout, K, s, p = case
    g = _g(hash(case) % 1000)
    x = torch.randn(N, Cin, Hh, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    ref = F.conv2d(q(x, H), q(w, H), stride=s, padding=p)
    y = ops.conv2d_nhwc(to_nhwc(x, H), ops.pack_weight(w.cuda(), H), Cout, K, K, s, p, tile=tile)
    d = err(from_nhwc(y)[:, :Cout], ref)
    report(f"conv_fp16[{case},t{tile}]", maxabs=d[0], rel=d[2])
    assert y.dtype == H and d[2] < 1e-3                     # one fp16 rounding of the output: 2^-11 relative


def test_conv_fp16_eval_epilogue_and_refusals():
    """Folded BN + residual + ReLU epilogue in fp16; the register-staged / prologue kernels and the training executor refuse fp16."""
    from tinyfaces import _hip, ops
    from tinyfaces.models.model import DetectionModel
    g = _g(3)
    N, Hh, W, Cin, Cout = 2, 13, 17, 512, 125
    x = torch.randn(N, Cin, Hh, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(N, 128, Hh, W, generator=g)
    ref = torch.relu(F.conv2d(q(x, H), q(w, H)) * sc[:Cout].view(1, -1, 1, 1) + sh[:Cout].view(1, -1, 1, 1) + q(res, H)[:, :Cout])
    y = ops.conv2d_nhwc(to_nhwc(x, H), ops.pack_weight(w.cuda(), H), Cout, 1, 1, 1, 0, ldy=128,
                        epi=_hip.EPI_AFFINE | _hip.EPI_RES | _hip.EPI_RELU, epi_scale=sc.cuda(), epi_shift=sh.cuda(), aux=to_nhwc(res, H))
    d = err(from_nhwc(y)[:, :Cout], ref)
    assert d[2] < 1e-3
    m = DetectionModel(num_templates=25).cuda().set_compute_dtype(H).train()
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):  # fp16 is an inference dtype here
        m(torch.zeros(1, 3, 64, 64, device="cuda"))


@pytest.mark.parametrize("ci", [0, 1])
def test_eval_forward_fp16_vs_reference_golden(golden, ci):
    """The whole detector in fp16 against the reference's own fp32 output: 10 mantissa bits -> tighter than the bf16 bar."""
    from tinyfaces.models.model import DetectionModel
    from oracle.model import OracleDetectionModel, tame_init_
    g = golden("model")
    m = DetectionModel(num_templates=25)
    m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
    m = m.cuda().set_compute_dtype(H).eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g[f"m{ci}_x"]).cuda()).cpu().numpy()
    d = err(y, g[f"m{ci}_y"])
    report(f"model_eval_fp16[{ci}]", maxabs=d[0], maxref=d[1])
    assert np.isfinite(y).all() and d[0] < 1.5e-3


def _template_boxes(rng, n, templates, width, height):
    """uniformly placed boxes of the template sizes, scores N(0,1)."""
    t = templates[rng.randint(0, templates.shape[0], n)]
    w, h = t[:, 2] - t[:, 0] + 1, t[:, 3] - t[:, 1] + 1
    cx, cy = rng.uniform(0, width, n), rng.uniform(0, height, n)
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1), rng.randn(n)


def test_nms_batched_is_index_exact_per_segment(templates):
    """tf_nms_f64_batched: ragged segments incl. empty, 1-box, non-multiples of 64, one above the 12 288-box prefetch limit; ties
    and duplicates inside a segment; every segment index-exact vs the restated torchvision kernel, and equal to the one-list call."""
    from tinyfaces import ops
    from oracle.nms import nms as onms
    rng = np.random.RandomState(5)
    sizes = [0, 1, 257, 4096, 0, 63, 20000, 64, 1251]
    boxes, scores = [], []
    for n in sizes:
        b, s = _template_boxes(rng, n, templates, 1500, 1100)
        if n >= 257:
            s[: n // 4] = np.round(s[: n // 4], 1)                 # many exact score ties
            b[5] = b[9]                                            # duplicate boxes
        boxes.append(b); scores.append(s)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    B, S = torch.from_numpy(np.concatenate(boxes)).cuda(), torch.from_numpy(np.concatenate(scores)).cuda()
    keeps = ops.nms_batched(B, S, offs.tolist(), 0.3)
    assert len(keeps) == len(sizes)
    for k, (n, b, s, o) in enumerate(zip(sizes, boxes, scores, offs)):
        want = onms(b, s, 0.3) + o if n else np.empty(0, np.int64)
        got = keeps[k].cpu().numpy()
        assert np.array_equal(got, want), f"segment {k} (n={n})"
        if n:
            single = ops.nms(B[o:o + n].contiguous(), S[o:o + n].contiguous(), 0.3).cpu().numpy()
            assert np.array_equal(single + o, got)
    report("nms_batched", segments=len(sizes), total=int(offs[-1]), kept=str([int(k.numel()) for k in keeps]))
    with pytest.raises(ValueError):
        ops.nms_batched(B, S, [0, 5, 3, int(offs[-1])], 0.3)


def test_nms_65536_boxes_vs_oracle(templates):
    """cfg5 size: N = 65 536 boxes of template sizes, scores N(0,1) (the bit matrix alone is 512 MiB).  The canvas is 2000 x 1500 so
    that most boxes are suppressed: the CPU oracle's cost is (kept boxes) x N."""
    from tinyfaces import ops
    from oracle.nms import nms as onms
    rng = np.random.RandomState(9)
    b, s = _template_boxes(rng, 65536, templates, 2000, 1500)
    keep = ops.nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), 0.3).cpu().numpy()
    want = onms(b, s, 0.3)
    report("nms_65536", kept=int(keep.size), ref_kept=int(want.size))
    assert np.array_equal(keep, want)
    # two such lists in one batched call
    b2, s2 = _template_boxes(rng, 20000, templates, 1200, 900)
    keeps = ops.nms_batched(torch.from_numpy(np.concatenate([b, b2])).cuda(), torch.from_numpy(np.concatenate([s, s2])).cuda(),
                            [0, 65536, 85536], 0.3)
    assert np.array_equal(keeps[0].cpu().numpy(), want) and np.array_equal(keeps[1].cpu().numpy() - 65536, onms(b2, s2, 0.3))


def test_top_level_3750x5000_fp32_vs_cpu_oracle():
    """The shapes that exist ONLY at the 5000-px level of configs[4] -- M = 293 125 pixels at /16, 1.17 M at /8, 4.7 M stem pixels,
    grids beyond 65 535 blocks, byte offsets beyond 2^32 inside the 25.7 GB fp32 arena -- against an INDEPENDENT implementation:
    ONE eval-mode forward of a 1 x 3 x 3750 x 5000 image through the CPU oracle (torch-CPU fp32 restatement of
    tinyfaces/models/model.py:89-128, ~5.3 TFLOP: about a minute on the box's host cores) and through the HIP fp32 path.
    north_star's bar: per-anchor cls / reg maps within 1e-3.  The fp16-vs-fp32 check of the next test is anchored by this one."""
    import os
    import time
    from tinyfaces.models.model import DetectionModel
    from oracle.model import OracleDetectionModel, tame_init_
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    om = tame_init_(OracleDetectionModel(num_templates=25), 0).eval()
    m = DetectionModel(num_templates=25)
    m.load_state_dict(om.state_dict(), strict=True)
    m = m.cuda().eval().set_compute_dtype(torch.float32)
    x = torch.randn(1, 3, 3750, 5000, generator=_g(5))
    with torch.no_grad():
        y = m(x.cuda()).cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = om(x)
        t_cpu = time.perf_counter() - t0
    assert y.shape == ref.shape == (1, 125, 469, 625)
    d_cls, d_reg = err(y[:, :25].numpy(), ref[:, :25].numpy()), err(y[:, 25:].numpy(), ref[:, 25:].numpy())
    report("hard_top_level_fp32_vs_oracle", cls_maxabs=d_cls[0], cls_maxref=d_cls[1], reg_maxabs=d_reg[0], reg_maxref=d_reg[1], oracle_cpu_s=t_cpu)
    assert torch.isfinite(y).all()
    assert d_cls[0] < 1e-3 and d_reg[0] < 1e-3


def test_pyramid_5000px_fp16_and_batched_detections(templates):
    """The hard setting end to end: a 1875 x 2500 image with scales (-1, 0, 1) -> levels 937x1250, 1875x2500, 3750x5000 (long side
    5000 px, one un-tiled forward: the activation arena of that level is asked from the executor and reported), fp16 operands.
    Checked against the fp32 HIP path on the SAME image (that path is pinned to the oracle at 960x1280 in test_gpu_fullsize.py):
    top-level maps within the fp16 bar, and nearly the same surviving boxes; then two images through get_detections_batch ==
    the per-image loop, row for row."""
    from tinyfaces import _hip, transforms
    from tinyfaces.evaluation import get_detections, get_detections_batch
    from tinyfaces.models.model import DetectionModel
    from oracle.model import OracleDetectionModel, tame_init_
    from oracle.targets import RF
    m = DetectionModel(num_templates=25)
    m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
    m = m.cuda().eval()
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    gen = _g(0)
    img = torch.rand(3, 1875, 2500, generator=gen)
    arena16 = _hip.lib().tf_detnet_workspace_bytes(_hip.TF_F16, 1, 3750, 5000, 125, 0)
    arena32 = _hip.lib().tf_detnet_workspace_bytes(_hip.TF_F32, 1, 3750, 5000, 125, 0)
    # top level alone, fp16 vs fp32
    x = torch.randn(1, 3, 3750, 5000, generator=gen).cuda()
    with torch.no_grad():
        y32 = m.set_compute_dtype(torch.float32)(x).cpu()
        torch.cuda.synchronize()
        y16 = m.set_compute_dtype(H)(x).cpu()
    d = err(y16.numpy(), y32.numpy())
    assert y16.shape == (1, 125, 469, 625) and torch.isfinite(y16).all()
    # calibrate a threshold that leaves a few thousand candidates (random weights): quantile of the fp32 top-level probabilities
    p = torch.sigmoid(y32[0, :25]).flatten()
    thr = float(torch.quantile(p[torch.randperm(p.numel(), generator=gen)[:2000000]], 0.9995))
    kw = dict(prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", pyramid_on_gpu=True)
    d32 = get_detections(m.set_compute_dtype(torch.float32), img, templates, RF, tf, **kw)
    d16 = get_detections(m.set_compute_dtype(H), img, templates, RF, tf, **kw)
    common = len(set(map(tuple, np.round(d16[:, :4], 0))) & set(map(tuple, np.round(d32[:, :4], 0))))
    report("hard_5000px", arena_fp16_gb=arena16 / 2**30, arena_fp32_gb=arena32 / 2**30, top_maxabs=d[0], top_maxref=d[1], thr=thr,
           kept_fp32=d32.shape[0], kept_fp16=d16.shape[0], common=common)
    assert d[0] < 3e-3                                            # fp16 vs fp32 maps at 3750 x 5000
    assert d32.shape[0] > 50 and common >= 0.9 * d32.shape[0] and abs(d16.shape[0] - d32.shape[0]) <= 0.1 * d32.shape[0]
    assert 10 * 2**30 < arena16 < 40 * 2**30                      # ~25 GB, no tiling needed in 288 GB
    # batched: two images, ONE nms call == the loop
    img2 = torch.rand(3, 700, 933, generator=gen)
    loop = [get_detections(m, im, templates, RF, tf, **kw) for im in (img, img2)]
    bat = get_detections_batch(m, [img, img2], templates, RF, tf, **{k: v for k, v in kw.items()})
    assert len(bat) == 2 and all(np.array_equal(a, b) for a, b in zip(loop, bat))
