"""-m gpu: HIP kernels (through the C ABI) vs the CPU oracle and the golden vectors:
dense_overlap targets, NMS, decode, criterion, SGD.  Integer / index / float64 work must be
bit-exact; float32 results within the tolerance written next to each assert."""
import numpy as np
import pytest
import torch

from gpu_util import err, report

pytestmark = pytest.mark.gpu


def _valid(b):
    return b[~np.logical_or(b[:, 2] <= b[:, 0], b[:, 3] <= b[:, 1])]


def _noise(g, tag, ngv):
    np.random.seed(int(g[f"{tag}_seed"]))
    return np.random.rand(63, 63, 25, ngv) if ngv else np.zeros((63, 63, 25, 0))


def test_probe_tr16(hip):
    """Record the lane mapping of ds_read_b64_tr_b16 that csrc/wgrad.hip relies on."""
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    assert hip.lib().tf_probe_tr16(out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    # assumed: within each 16-lane group, lane i supplies row (i>>2), cols 4*(i&3).. of a [4][16] block and
    # receives column i:  result[lane][j] = element supplied by lane (16*grp + 4*j + (i>>2)), position (i&3)
    exp = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        grp, i = lane // 16, lane % 16
        for j in range(4):
            src_lane = 16 * grp + 4 * j + (i >> 2)
            exp[lane, j] = src_lane * 4 + (i & 3)
    ok = np.array_equal(got, exp)
    report("probe_tr16", ok=ok, lane0=str(got[0].tolist()), lane1=str(got[1].tolist()), lane5=str(got[5].tolist()),
           lane17=str(got[17].tolist()), lane63=str(got[63].tolist()))
    import os
    from gpu_util import OUT
    np.savetxt(os.path.join(OUT, "tr16_probe.txt"), got, fmt="%d")
    assert ok, got[:20]


@pytest.mark.parametrize("kind", [7, 8])
def test_grid_barrier_probe_is_coherent(hip, kind):
    """csrc/probe.hip kinds 7 / 8 (scripts/gridbar.py): 40 device-wide barriers inside one launch of 256 co-resident blocks, every
    thread writing 16 bytes before and reading another block's 16 bytes after each barrier.  The measurement is only worth quoting if the
    barrier IS one: no spin gave up, no read saw a value older than the current round, every block arrived at every barrier."""
    blocks, iters = 256, 40
    buf = torch.zeros(4096 + blocks * 256 * 16 + 4096, dtype=torch.uint8, device="cuda")
    rc = hip.lib().tf_debug_probe(kind, blocks, 0, buf.data_ptr(), buf.numel(), iters, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    flags = buf[:16].view(torch.int32).tolist()
    report(f"grid_barrier_probe[{kind}]", arrivals=flags[0], gave_up=flags[1], stale_reads=flags[2])
    assert flags[1] == 0 and flags[2] == 0
    assert flags[0] == (blocks * iters if kind == 7 else 8 * iters)
    assert hip.lib().tf_debug_probe(kind, 2048, 0, buf.data_ptr(), buf.numel(), 1, torch.cuda.current_stream().cuda_stream) != 0   # more blocks than can be resident: refused


@pytest.mark.parametrize("ci", range(5))
def test_targets_vs_reference_golden(golden, ci):
    from tinyfaces import ops
    from oracle import targets as otgt
    g = golden("targets")
    t = g["templates"]
    tag = f"c{ci}"
    boxes = g[f"{tag}_boxes"]
    bv = _valid(boxes)
    noise = _noise(g, tag, bv.shape[0])
    iou = ops.dense_overlap_iou(bv, t).cpu().numpy()
    ref_iou = otgt.dense_overlap(-1, -1, 8, 8, 63, 63, t[:, 0], t[:, 1], t[:, 2], t[:, 3], bv[:, 0], bv[:, 1], bv[:, 2], bv[:, 3])
    iou_exact = np.array_equal(iou, ref_iou)
    cm, rm = ops.dense_overlap_targets([boxes], t, paste_boxes=[g[f"{tag}_paste"]], flips=[int(g[f"{tag}_flip"])], noise=[noise])
    cm = cm.cpu().numpy()[0].transpose(1, 2, 0)          # CHW -> HWC like the fixture
    rm = rm.cpu().numpy()[0].transpose(1, 2, 0)
    pad = otgt.get_padding(t, g[f"{tag}_paste"])
    if int(g[f"{tag}_flip"]):
        pad = np.fliplr(pad)
    ocm, orm, _ = otgt.get_heatmaps(boxes.copy(), t, pad, noise=noise)
    cls_mismatch = int((cm != ocm).sum())
    rm_ref = orm.astype(np.float32)
    reg_mismatch = int((rm != rm_ref).sum())
    d, m, _ = err(rm, rm_ref)
    report(f"targets[{ci}]", iou_exact=iou_exact, cls_mismatch=cls_mismatch, reg_mismatch=reg_mismatch, reg_maxabs=d)
    assert iou_exact                                         # float64, bit-exact
    assert np.array_equal(cm.astype(np.int8), g[f"{tag}_class"])       # labels identical to the reference
    # tx/ty are exact; tw/th go through log() whose last f64 bit may differ between libm and the GPU, which
    # survives the cast to f32 only at a rounding boundary: allow <= 1 f32 ulp on <= 1e-5 of the entries
    assert reg_mismatch <= max(1, rm.size // 100000)
    assert np.allclose(rm, rm_ref, rtol=2e-7, atol=0)


def test_targets_rng_mode_properties(golden):
    from tinyfaces import ops
    t = golden("targets")["templates"]
    rng = np.random.RandomState(0)
    boxes = []
    for _ in range(12):
        gcount = rng.randint(1, 17)
        w = np.exp(rng.uniform(np.log(8), np.log(200), gcount)); h = w * rng.uniform(1, 1.5, gcount)
        x1 = rng.uniform(0, 500 - w); y1 = rng.uniform(0, 500 - h)
        boxes.append(np.stack([x1, y1, x1 + w, y1 + h], 1))
    cm1, rm1 = ops.dense_overlap_targets(boxes, t, paste_boxes=[[0, 0, 500, 500]] * 12, seed=5)
    cm2, rm2 = ops.dense_overlap_targets(boxes, t, paste_boxes=[[0, 0, 500, 500]] * 12, seed=5)
    assert torch.equal(cm1, cm2) and torch.equal(rm1, rm2)                # deterministic per seed
    assert set(torch.unique(cm1).tolist()) <= {-1.0, 0.0, 1.0}
    # with the tie-break noise <= 1e-6 the labels equal the oracle's except where IoU is within 1e-6 of a threshold
    from oracle import targets as otgt
    pad = otgt.get_padding(t, [0, 0, 500, 500])
    diff = 0
    for b in range(12):
        ocm, _, _ = otgt.get_heatmaps(boxes[b].copy(), t, pad, noise=np.zeros((63, 63, 25, boxes[b].shape[0])))
        diff += int((cm1[b].cpu().numpy().transpose(1, 2, 0) != ocm).sum())
    report("targets_rng", label_diff_vs_noiseless_oracle=diff, positives=int((cm1 == 1).sum()))
    assert diff <= 25
    assert int((cm1 == 1).sum()) > 12


@pytest.mark.parametrize("ci", range(5))
def test_nms_golden(golden, ci):
    from tinyfaces import ops
    g = golden("nms")
    tag = f"n{ci}"
    b = torch.from_numpy(g[f"{tag}_boxes"]).reshape(-1, 4).cuda()
    s = torch.from_numpy(g[f"{tag}_scores"]).cuda()
    keep = ops.nms(b, s, 0.3).cpu().numpy()
    report(f"nms[{ci}]", n=b.shape[0], k=keep.size, exact=bool(np.array_equal(keep, g[f"{tag}_keep"])))
    assert np.array_equal(keep, g[f"{tag}_keep"])                       # identical indices, identical order


@pytest.mark.parametrize("n,seed", [(63, 1), (64, 2), (65, 3), (1000, 4), (5000, 5), (12288, 7), (12289, 8), (20000, 6)])
def test_nms_random_vs_oracle(n, seed):
    from tinyfaces import ops
    from oracle.nms import nms as onms
    rng = np.random.RandomState(seed)
    cx, cy = rng.uniform(0, 1500, n), rng.uniform(0, 1000, n)
    w = np.exp(rng.uniform(np.log(8), np.log(150), n)); h = w * rng.uniform(1.0, 1.4, n)
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    scores = np.round(rng.randn(n), 2)                                  # many exact ties -> exercises stability
    keep = ops.nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 0.3).cpu().numpy()
    ref = onms(boxes, scores, 0.3)
    report(f"nms_random[{n}]", k=keep.size, kref=ref.size, exact=bool(np.array_equal(keep, ref)))
    assert np.array_equal(keep, ref)


@pytest.mark.parametrize("ci", range(4))
def test_decode_golden(golden, ci):
    from tinyfaces.models.utils import get_bboxes
    from oracle.targets import RF
    g = golden("decode")
    t = golden("targets")["templates"]
    tag = f"d{ci}"
    b, s = get_bboxes(g[f"{tag}_score_cls"], g[f"{tag}_score_reg"], g[f"{tag}_prob"].copy(), t, float(g[f"{tag}_thr"]), RF,
                      float(g[f"{tag}_scale"]))
    rb, rs = g[f"{tag}_boxes"], g[f"{tag}_scores"]
    same_n = b.shape[0] == rb.shape[0]
    d = err(b, rb)[0] if same_n else -1
    report(f"decode[{ci}]", n=b.shape[0], nref=rb.shape[0], box_maxabs=d)
    assert same_n                                                       # same candidates ...
    assert np.array_equal(s, rs)                                        # ... in the same order (scores are the raw logits)
    # boxes: np.exp of a float32 array is numpy's own f32 routine (faithful, not always correctly rounded); the
    # kernel rounds an f64 exp to f32.  They differ by <= 1 f32 ulp of exp(), i.e. <= 1.2e-7 * (box size / scale).
    size = float(max((rb[:, 2] - rb[:, 0]).max(), (rb[:, 3] - rb[:, 1]).max()))
    assert np.abs(b - rb).max() <= 1.3e-7 * size
    # refine=False (utils.py:65-66): the reference returns the x1 ROW of the anchor boxes (`bboxes[0]` of a (4, N) array): same values
    # exactly (no exp involved), same candidates; unrefined="boxes" gives the (N, 4) anchors whose first column that row is
    b0, s0 = get_bboxes(g[f"{tag}_score_cls"], g[f"{tag}_score_reg"], g[f"{tag}_prob"].copy(), t, float(g[f"{tag}_thr"]), RF,
                        float(g[f"{tag}_scale"]), refine=False)
    assert b0.shape == g[f"{tag}_norefine"].shape and np.array_equal(b0, g[f"{tag}_norefine"]) and np.array_equal(s0, g[f"{tag}_norefine_scores"])
    b4, _ = get_bboxes(g[f"{tag}_score_cls"], g[f"{tag}_score_reg"], g[f"{tag}_prob"].copy(), t, float(g[f"{tag}_thr"]), RF,
                       float(g[f"{tag}_scale"]), refine=False, unrefined="boxes")
    assert b4.shape == rb.shape and np.array_equal(b4[:, 0], b0) and np.all(b4[:, 2] > b4[:, 0])


def test_decode_w_lt_25_raises_like_reference(golden):
    from tinyfaces.models.utils import get_bboxes
    from oracle.targets import RF
    z = np.zeros((1, 5, 20, 25), np.float32)
    with pytest.raises(IndexError):
        get_bboxes(z, np.zeros((1, 5, 20, 100), np.float32), z.copy(), golden("targets")["templates"], 0.5, RF, 1)


def _keep_flags(records, B, E):
    pk = np.ones((B, E), np.uint8); nk = np.ones((B, E), np.uint8)
    for b, r in enumerate(records):
        pk[b, :r["pos_keep"].size] = r["pos_keep"]
        nk[b, :r["neg_keep"].size] = r["neg_keep"]
    return torch.from_numpy(pk), torch.from_numpy(nk)


@pytest.mark.parametrize("ci", range(4))
def test_criterion_golden(golden, ci):
    from tinyfaces import ops
    from oracle import criterion as ocrit
    g = golden("criterion")
    tag = f"k{ci}"
    out = torch.from_numpy(g[f"{tag}_output"])
    cm = torch.from_numpy(g[f"{tag}_class_map"].astype(np.float32))
    rm = torch.from_numpy(g[f"{tag}_reg_map"])
    np.random.seed(int(g[f"{tag}_seed"]))
    r = ocrit.criterion(out, cm, rm)                                    # replays the reference's permutation draws
    B, _, H, W = out.shape
    pk, nk = _keep_flags(r["records"], B, 25 * H * W)
    cm_d = cm.clone().cuda()
    loss2, grad, labels = ops.criterion_fwd_bwd(out.cuda(), cm_d, rm.cuda(), pos_keep=pk, neg_keep=nk, want_labels=True)
    loss2 = loss2.cpu().numpy()
    ref = g[f"{tag}_loss"]
    lab_mis = int((labels.cpu() != r["class_map_final"]).sum())
    # loss.py:62 mines the caller's tensor in place.  (On a CPU tensor the reference's balance_sample ALSO writes
    # through .cpu().numpy() aliasing, which is what the fixture recorded; on a device tensor only OHEM does.)
    mined = cm.clone()
    mined[torch.nn.functional.soft_margin_loss(out[:, :25], cm, reduction="none") < 0.03] = 0
    ohem_mis = int((cm_d.cpu() != mined).sum())
    assert np.array_equal(r["class_map_final"].numpy().astype(np.int8), g[f"{tag}_class_after_ohem_inplace"])
    gd = err(grad.cpu().numpy(), g[f"{tag}_grad"])
    report(f"criterion[{ci}]", cls=loss2[0], cls_ref=ref[1], reg=loss2[1], reg_ref=ref[2], label_mismatch=lab_mis,
           ohem_mismatch=ohem_mis, grad_maxabs=gd[0])
    assert ohem_mis == 0 and lab_mis == 0                               # same mined + sampled labels as the reference
    assert np.allclose([loss2[0], loss2[1]], [ref[1], ref[2]], rtol=1e-5)      # fp32 sums, different summation order
    assert np.allclose(grad.cpu().numpy(), g[f"{tag}_grad"], rtol=1e-5, atol=1e-6)


def test_criterion_rng_mode(golden):
    from tinyfaces import ops
    g = torch.Generator().manual_seed(0)
    out = (torch.randn(3, 125, 63, 63, generator=g) * 1.5).cuda()
    cm0 = torch.zeros(3, 25 * 63 * 63)
    for b in range(3):
        perm = torch.randperm(cm0.shape[1], generator=g)
        cm0[b, perm[:400]] = 1
        cm0[b, perm[400:60000]] = -1
    cm0 = cm0.view(3, 25, 63, 63)
    rm = torch.randn(3, 100, 63, 63, generator=g).cuda()
    l1, g1, lab1 = ops.criterion_fwd_bwd(out, cm0.clone().cuda(), rm, seed=11, want_labels=True)
    l2, g2, lab2 = ops.criterion_fwd_bwd(out, cm0.clone().cuda(), rm, seed=11, want_labels=True)
    l3, g3, lab3 = ops.criterion_fwd_bwd(out, cm0.clone().cuda(), rm, seed=12, want_labels=True)
    assert torch.equal(lab1, lab2) and torch.equal(g1, g2)              # deterministic per seed
    assert not torch.equal(lab1, lab3)
    for b in range(3):
        npos, nneg = int((lab1[b] == 1).sum()), int((lab1[b] == -1).sum())
        assert npos == 128 and nneg == 128, (npos, nneg)                 # utils.py:119-137 caps, enough candidates
    # non-zero gradient only where a label survived
    assert int((g1[:, :25] != 0).sum()) <= 3 * 256
    report("criterion_rng", cls=float(l1[0]), reg=float(l1[1]))


def test_sgd_vs_torch():
    from tinyfaces import ops
    g = torch.Generator().manual_seed(1)
    for n in (1, 5, 1023, 4096 + 3, 1 << 20):
        p0 = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) for _ in range(3)]
        pt = p0.clone().requires_grad_(True)
        opt = torch.optim.SGD([pt], lr=0.01, momentum=0.9, weight_decay=5e-4)
        pd, md = p0.clone().cuda(), torch.zeros(n).cuda()
        for gr in grads:
            pt.grad = gr.clone()
            opt.step()
            ops.sgd_step(pd, gr.cuda(), md, 0.01, 0.9, 5e-4)
        d = err(pd.cpu().numpy(), pt.detach().numpy())
        report(f"sgd[{n}]", maxabs=d[0])
        assert np.allclose(pd.cpu().numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-7)
    # unaligned segment of a flat buffer
    flat = torch.randn(1000).cuda(); gflat = torch.randn(1000).cuda(); m = torch.zeros(1000).cuda()
    ref = flat[3:500].cpu() - 0.1 * (gflat[3:500].cpu() + 0.0 * flat[3:500].cpu())
    ops.sgd_step(flat[3:500], gflat[3:500], m[3:500], 0.1, 0.0, 0.0)
    assert np.allclose(flat[3:500].cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


def test_pairwise_iou_distance_and_kmedoids_vs_reference_golden(golden):
    """the n^2 distance matrix of the template clustering on the device is BIT-equal to the reference's
    compute_distances (tests/golden/clustering.npz holds its output), so the reference's own k-medoids run on it picks exactly
    the golden medoids; plus a 3000-box matrix against the vectorised numpy expression."""
    from tinyfaces import clustering as cl
    g = golden("clustering")
    shapes = cl.centralize_bbox(g["boxes"])
    dist = cl.compute_distances(shapes, device="cuda")
    assert dist.dtype == np.float64 and np.array_equal(dist, g["dist"])
    for k in (3, 7):
        med, member = cl.k_medoids(dist, k, rng=np.random.RandomState(40 + k))
        assert np.array_equal(med, g[f"k{k}_medoids"]) and np.array_equal(member, g[f"k{k}_member"])
    rng = np.random.RandomState(1)
    wh = np.exp(rng.uniform(np.log(4), np.log(400), (3000, 2)))
    big = np.concatenate([-wh / 2, wh / 2], 1)
    big[7] = 0.0                                              # a degenerate box: union <= 0 against itself -> IoU 0, distance 1
    d = cl.compute_distances(big, device="cuda")
    assert np.array_equal(d, cl.compute_distances(big)) and d[7, 7] == 1.0
    res = cl.compute_kmedoids(g["boxes"], 1, option="local", indices=4, max_clusters=5, rng=np.random.RandomState(0), device="cuda")
    ref = cl.compute_kmedoids(g["boxes"], 1, option="local", indices=4, max_clusters=5, rng=np.random.RandomState(0))
    assert all(np.array_equal(np.array(a["medoids"]), np.array(b["medoids"])) for a, b in zip(res[4:], ref[4:]))
