"""Ad-hoc GPU diagnostics (not part of the test-suite): per-parameter gradient error profile of the
executor vs the CPU oracle, and NMS stage dumps."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd"), os.path.join(ROOT, "tests")]
from oracle.model import OracleDetectionModel, tame_init_
from oracle.nms import nms as onms
from tinyfaces.models.model import DetectionModel
from tinyfaces import ops, _hip

def grad_profile(dtype, shape=(2, 3, 96, 128), dense=False):
    """per-tensor errors of the HIP executor and of torch-fp32, both measured against a float64 oracle"""
    def oracle(dt):
        om = tame_init_(OracleDetectionModel(num_templates=25), 0).train().to(dt)
        g = torch.Generator().manual_seed(52)
        x = torch.randn(*shape, generator=g)
        yo = om(x.to(dt))
        gy = torch.randn(yo.shape, generator=g) * ((torch.rand(yo.shape, generator=g) < 0.02) if not dense else 1.0)
        yo.backward(gy.to(dt))
        return om, x, gy, yo
    o64, x, gy, y64 = oracle(torch.float64)
    o32, _, _, y32 = oracle(torch.float32)
    m = DetectionModel(num_templates=25); m.load_state_dict(o32.state_dict())
    m = m.cuda().set_compute_dtype(dtype).train()
    y = m(x.cuda()); y.backward(gy.cuda())
    p64, p32 = dict(o64.named_parameters()), dict(o32.named_parameters())
    rh, rt, cs = [], [], []
    for k, p in m.named_parameters():
        if p.grad is None or k.startswith("score4_upsample"): continue
        ref = p64[k].grad
        a = p.grad.cpu().double()
        rh.append(float((a - ref).abs().max() / (ref.abs().max() + 1e-30)))
        rt.append(float((p32[k].grad.double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
        cs.append(float((a * ref).sum() / (a.norm() * ref.norm() + 1e-30)))
    rh, rt, cs = np.array(rh), np.array(rt), np.array(cs)
    q = lambda v: [float(np.quantile(v, t)) for t in (0.5, 0.9, 1.0)]
    print(f"== grad profile {dtype} {shape} dense={dense}: y maxabs vs f64 {float((y.detach().cpu().double()-y64.detach()).abs().max()):.3e} "
          f"(torch32: {float((y32.detach().double()-y64.detach()).abs().max()):.3e})")
    print("   rel-to-max err  hip   [median, p90, max]:", ["%.2e" % v for v in q(rh)])
    print("   rel-to-max err  torch32 vs f64          :", ["%.2e" % v for v in q(rt)])
    print("   cosine(hip, f64) [min, p10, median]:", "%.5f %.5f %.5f" % (cs.min(), np.quantile(cs, 0.1), np.median(cs)))

def nms_debug(n, seed, boxes=None, scores=None):
    if boxes is None:
        rng = np.random.RandomState(seed)
        cx, cy = rng.uniform(0, 1500, n), rng.uniform(0, 1000, n)
        w = np.exp(rng.uniform(np.log(8), np.log(150), n)); h = w * rng.uniform(1.0, 1.4, n)
        boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        scores = np.round(rng.randn(n), 2)
    bd, sd = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    l = _hip.lib()
    wsb = l.tf_nms_workspace_bytes(n)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    keep = torch.empty(n, dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = l.tf_nms_f64(bd.data_ptr(), sd.data_ptr(), n, 0.3, keep.data_ptr(), cnt.data_ptr(), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a256 = lambda v: (v + 255) & ~255
    order = ws[:n * 4].view(torch.int32).cpu().numpy()
    nwords = (n + 63) // 64
    off = a256(n * 4) + a256(n * 32)
    mask = ws[off: off + n * nwords * 8].view(torch.int64).cpu().numpy().view(np.uint64).reshape(n, nwords)
    ref_order = np.argsort(-scores, kind="stable")
    print(f"== nms n={n}: rc={rc} k={int(cnt)} ref_k={onms(boxes, scores, 0.3).size} order_ok={np.array_equal(order, ref_order)}")
    sb = boxes[ref_order]
    area = (sb[:, 2] - sb[:, 0]) * (sb[:, 3] - sb[:, 1])
    bad = 0
    for i in range(n if n <= 5000 else 0):
        xx1 = np.maximum(sb[i, 0], sb[:, 0]); yy1 = np.maximum(sb[i, 1], sb[:, 1]); xx2 = np.minimum(sb[i, 2], sb[:, 2]); yy2 = np.minimum(sb[i, 3], sb[:, 3])
        inter = np.maximum(0, xx2 - xx1) * np.maximum(0, yy2 - yy1)
        ovr = inter / (area[i] + area - inter)
        bits = (ovr > 0.3) & (np.arange(n) > i)
        for wd in range(i // 64, nwords):
            exp = 0
            for j in np.nonzero(bits[wd * 64:(wd + 1) * 64])[0]: exp |= (1 << int(j))
            if int(mask[i, wd]) != exp:
                bad += 1
                if bad < 5: print("  mask mismatch row", i, "word", wd, hex(int(mask[i, wd])), hex(exp))
    print("  mask mismatches:", bad)
    # emulate the scan on the host from the device mask
    removed = np.zeros(nwords, dtype=np.uint64); k = []
    for i in range(n):
        if not (int(removed[i // 64]) >> (i % 64)) & 1:
            k.append(order[i]); removed |= mask[i] if True else 0
    print("  host scan over device mask ->", len(k), " device keep ->", int(cnt), " same:", np.array_equal(np.array(k), keep[:int(cnt)].cpu().numpy()))

def amp_profile(shape=(2, 3, 256, 256)):
    """Calibration only: what torch's own bf16 autocast (MIOpen kernels) achieves vs a float64 oracle on the same problem."""
    def oracle(dt, dev, amp=False):
        om = tame_init_(OracleDetectionModel(num_templates=25), 0).train().to(dt).to(dev)
        g = torch.Generator().manual_seed(52)
        x = torch.randn(*shape, generator=g)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            yo = om(x.to(dt).to(dev))
        gy = torch.randn(yo.shape, generator=g)
        yo.float().backward(gy.to(yo.device).float() if amp else gy.to(dt).to(dev))
        return om, yo
    o64, y64 = oracle(torch.float64, "cpu")
    oamp, yamp = oracle(torch.float32, "cuda", amp=True)
    p64 = dict(o64.named_parameters())
    cs, rh = [], []
    for k, p in oamp.named_parameters():
        if p.grad is None or k.startswith("score4_upsample"): continue
        ref = p64[k].grad; a = p.grad.cpu().double()
        cs.append(float((a * ref).sum() / (a.norm() * ref.norm() + 1e-30)))
        rh.append(float((a - ref).abs().max() / (ref.abs().max() + 1e-30)))
    cs, rh = np.array(cs), np.array(rh)
    print(f"== torch autocast(bf16) on GPU {shape}: y maxabs vs f64 {float((yamp.detach().cpu().double()-y64.detach()).abs().max()):.3e}")
    print("   rel-to-max err [median, p90, max]:", ["%.2e" % float(np.quantile(rh, t)) for t in (0.5, 0.9, 1.0)])
    print("   cosine [min, p10, median]: %.5f %.5f %.5f" % (cs.min(), np.quantile(cs, 0.1), np.median(cs)))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("nms", "all"):
        for n, s in ((65, 3), (128, 1), (130, 2), (4096, 7), (20000, 6)):
            nms_debug(n, s)
        g = np.load(os.path.join(ROOT, "tests/golden/nms.npz"))
        nms_debug(4096, 0, g["n4_boxes"], g["n4_scores"])
    if which in ("amp",):
        amp_profile()
        grad_profile(torch.bfloat16, (2, 3, 256, 256), dense=True)
    if which in ("grad", "all"):
        grad_profile(torch.float32)
        grad_profile(torch.bfloat16)
        grad_profile(torch.float32, (2, 3, 256, 256), dense=True)
        grad_profile(torch.bfloat16, (2, 3, 256, 256), dense=True)
