"""-m gpu: the whole detector through the reference call surface (DetectionModel, DetectionCriterion,
get_detections, trainer.train) vs the golden vectors produced by the reference's own source and
vs the CPU oracle.  Tolerance from BASELINE.json north_star: cls/reg maps within 1e-3 in fp32;
identical NMS-surviving boxes on fixed-seed inputs.  The bf16 fast path is held to 5e-3 on eval maps (measured 2.2e-3)
and 3e-2 on training-mode maps of the tiny fixtures (batch statistics over <= 400 pixels amplify the operand rounding;
measured 1.3e-2).  The benchmarked sizes are checked in tests/test_gpu_fullsize.py."""
import io
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from gpu_util import err, report

pytestmark = pytest.mark.gpu


def _load_oracle_weights(model, seed=0):
    """Same tamed init as the fixtures (oracle/model.py:tame_init_) -> load into the product model."""
    from oracle.model import OracleDetectionModel, tame_init_
    om = tame_init_(OracleDetectionModel(num_templates=25), seed)
    missing = model.load_state_dict(om.state_dict(), strict=True)
    return om


@pytest.fixture(scope="module")
def models():
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    om = _load_oracle_weights(m)
    return m.cuda(), om


def test_state_dict_contract(golden):
    from tinyfaces.models.model import DetectionModel
    g = golden("model")
    m = DetectionModel(num_objects=1, num_templates=25)
    sd = m.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"].tolist()
    assert [v.numel() for v in sd.values()] == g["state_dict_numel"].tolist()
    lp = m.learnable_parameters(0.5)
    assert [len(list(x["params"])) for x in lp] == [284, 2, 2, 1]
    assert [x["lr"] for x in lp] == g["group_lrs"].tolist()
    assert np.array_equal(m.score4_upsample.weight[3, 3].detach().numpy(), g["bilinear_diag"])


def test_cpu_input_raises():
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 3, 64, 64))


@pytest.mark.parametrize("ci", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_forward_vs_reference_golden(golden, models, ci, dtype):
    m, _ = models
    g = golden("model")
    tag = f"m{ci}"
    m.set_compute_dtype(dtype).eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g[f"{tag}_x"]).cuda()).cpu().numpy()
    d = err(y, g[f"{tag}_y"])
    report(f"model_eval[{ci},{dtype}]", maxabs=d[0], maxref=d[1])
    assert y.shape == g[f"{tag}_y"].shape
    assert d[0] < (1e-3 if dtype == torch.float32 else 5e-3)


def _oracle_train_pass(dtype_o, x, gy):
    from oracle.model import OracleDetectionModel, tame_init_
    om = tame_init_(OracleDetectionModel(num_templates=25), 0).train().to(dtype_o)
    y = om(x.to(dtype_o))
    y.backward(gy.to(dtype_o))
    return om, y.detach()


@pytest.mark.parametrize("ci", [2, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_forward_backward_vs_reference_golden(golden, ci, dtype):
    """Training-mode forward (batch-stat BN) and the full backward.

    Forward: within 1e-3 of the reference's own output (golden).  Gradients: a 100-layer ReLU network at this tiny
    size is chaotic at ReLU boundaries -- torch-fp32 itself is up to 1e-1 (relative to a tensor's max) away from a
    float64 evaluation, at different places than any other fp32 implementation.  So the gradient bar is stated
    against a float64 run of the oracle: the HIP fp32 path must be as close to it as torch-fp32 is (same order of
    magnitude at the median / 90th percentile over all 286 parameter tensors), the ReLU-free head gradients must
    match the reference golden to 1e-4, and the bf16 path must keep cosine similarity >= 0.9 with float64 for every
    tensor (torch's own bf16 autocast measures 0.93 min / 0.97 median on this problem; scripts/debug_gpu.py amp)."""
    from tinyfaces.models.model import DetectionModel
    g = golden("model")
    tag = f"m{ci}"
    x, gy = torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_gy"])
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(dtype).train()
    y = m(x.cuda())
    dy = err(y.detach().cpu().numpy(), g[f"{tag}_y"])
    y.backward(gy.cuda())
    o64, _ = _oracle_train_pass(torch.float64, x, gy)
    o32, _ = _oracle_train_pass(torch.float32, x, gy)
    p64, p32, params = dict(o64.named_parameters()), dict(o32.named_parameters()), dict(m.named_parameters())
    rh, rt, cs = [], [], []
    for k, p in params.items():
        if p.grad is None or k.startswith("score4_upsample"):
            continue
        ref = p64[k].grad
        a = p.grad.cpu().double()
        rh.append(float((a - ref).abs().max() / (ref.abs().max() + 1e-30)))
        rt.append(float((p32[k].grad.double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
        cs.append(float((a * ref).sum() / (a.norm() * ref.norm() + 1e-30)))
    rh, rt, cs = np.array(rh), np.array(rt), np.array(cs)
    head = max(err(params[k].grad.cpu().numpy(), g[f"{tag}_grad::{k}"])[2] for k in ("score_res4.bias",)
               if f"{tag}_grad::{k}" in g.files)
    # every parameter gradient the reference fixture stores (torch-fp32 of the reference's own model.py): relative error and
    # cosine against it.  (The fixture's `gx` has no counterpart: the training step never asks for the input gradient,
    # trainer.py:72-87 feeds a tensor that does not require grad, and conv1 computes no data gradient here.)
    gold = {k.split("::")[1]: g[k] for k in g.files if k.startswith(f"{tag}_grad::")}
    g_rel, g_cos = {}, {}
    for k, ref_g in gold.items():
        a = params[k].grad.cpu().double().numpy().ravel()
        b = ref_g.astype(np.float64).ravel()
        g_rel[k] = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
        g_cos[k] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    sd = m.state_dict()
    drm = err(sd["model.bn1.running_mean"].cpu().numpy(), g[f"{tag}_rm::model.bn1.running_mean"])
    drv = err(sd["model.layer3.5.bn2.running_var"].cpu().numpy(), g[f"{tag}_rv::model.layer3.5.bn2.running_var"])
    report(f"model_train[{ci},{dtype}]", y_maxabs=dy[0], hip_med=np.median(rh), hip_p90=np.quantile(rh, .9), hip_max=rh.max(),
           t32_med=np.median(rt), t32_p90=np.quantile(rt, .9), t32_max=rt.max(), cos_min=cs.min(), cos_med=np.median(cs),
           head=head, rm=drm[0], rv=drv[0], golden_rel=str({k: round(v, 6) for k, v in g_rel.items()}),
           golden_cos_min=min(g_cos.values()))
    assert len(rh) == 286 and len(gold) >= 3
    if dtype == torch.float32:
        assert dy[0] < 1e-3                                               # north_star: maps within 1e-3 in fp32
        assert head < 1e-4
        assert min(g_cos.values()) > 0.999, g_cos                         # all golden gradients, not one
        for k, v in g_rel.items():
            # a stored tensor may sit behind ReLU boundaries (see the docstring): bounded by what torch-fp32 itself does there
            t32 = float((p32[k].grad.double() - p64[k].grad).abs().max() / (p64[k].grad.abs().max() + 1e-30))
            assert v <= 4 * t32 + 2e-3, (k, v, t32)
        assert np.median(rh) <= 4 * np.median(rt) + 1e-4 and np.quantile(rh, .9) <= 4 * np.quantile(rt, .9) + 1e-3
        assert cs.min() > 0.999
        assert drm[0] < 1e-4 and drv[0] < 1e-3
    else:
        assert dy[0] < 3e-2 and head < 5e-3
        assert min(g_cos.values()) > 0.9, g_cos
        assert cs.min() > 0.9 and np.median(cs) > 0.95
        assert drm[0] < 5e-3 and drv[0] < 2e-2
    assert int(sd["model.bn1.num_batches_tracked"]) == 1
    assert params["model.fc.weight"].grad is None                       # dead fc keeps grad None like the reference (D4)


def test_get_detections_vs_reference_golden(golden, models):
    """End-to-end pyramid + decode + NMS (evaluation.py:20-87) on the golden image: same boxes, same order."""
    from tinyfaces import transforms
    from tinyfaces.evaluation import get_detections
    from oracle.targets import RF
    m, _ = models
    g = golden("detections")
    templates = golden("targets")["templates"]
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    m.set_compute_dtype(torch.float32)
    from oracle.nms import nms as onms
    dets, cand, keep = get_detections(m, torch.from_numpy(g["img"]), templates, RF, tf, prob_thresh=float(g["thr"]), nms_thresh=0.3,
                                      scales=tuple(g["scales"].tolist()), device="cuda", return_candidates=True)
    ref = g["dets_ref_K4"]
    same = dets.shape[0] == ref.shape[0]
    d = err(dets[:, :4], ref)[0] if same else -1
    report("get_detections", k=dets.shape[0], kref=ref.shape[0], maxabs=d, candidates=cand.shape[0])
    # north_star: identical NMS-surviving boxes -- same count, same order, same rows as the reference's own output ...
    assert dets.shape == (ref.shape[0], 5)
    assert np.allclose(dets[:, :4], ref, rtol=0, atol=1e-3), d
    # ... and the surviving INDICES are those of the restated torchvision kernel on the same candidate list
    assert np.array_equal(keep, onms(cand[:, :4], cand[:, 4], 0.3))
    assert np.array_equal(dets, cand[keep])


def test_get_detections_pyramid_on_gpu_is_identical(golden, models):
    """the pyramid levels built on the device (tf_image_prepare) are bit-identical to PIL + ToTensor + Normalize
    on the host, so the detections are exactly the same rows; a transform the fast path cannot mirror is refused loudly."""
    from tinyfaces import transforms
    from tinyfaces.evaluation import get_detections
    from oracle.targets import RF
    m, _ = models
    g = golden("detections")
    templates = golden("targets")["templates"]
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    m.set_compute_dtype(torch.bfloat16)
    kw = dict(prob_thresh=float(g["thr"]), nms_thresh=0.3, scales=tuple(g["scales"].tolist()), device="cuda")
    a = get_detections(m, torch.from_numpy(g["img"]), templates, RF, tf, **kw)
    b = get_detections(m, torch.from_numpy(g["img"]), templates, RF, tf, pyramid_on_gpu=True, **kw)
    report("get_detections_gpu_pyramid", k=a.shape[0], identical=bool(np.array_equal(a, b)))
    assert a.shape[0] > 0 and np.array_equal(a, b)
    with pytest.raises(ValueError, match="pyramid_on_gpu"):
        get_detections(m, torch.from_numpy(g["img"]), templates, RF, lambda im: transforms.ToTensor()(im), pyramid_on_gpu=True, **kw)


def test_get_detections_inside_a_session_repacks_nothing(golden, models):
    """r6: evaluate_model.py calls get_detections once per image inside ONE constant_weights() session.  get_detections used to call
    model.to(device) unconditionally; nn.Module.to walks `_apply` even as a no-op, DetectionModel._apply must then assume that every storage
    moved, and the pointer tables + the packed evaluation weights were rebuilt for every image (80 of 87 ms end to end at 1280 x 960).
    Now a model that already is on the device is left alone: the table key and the packed-weights key survive the calls, and a model that IS
    elsewhere still gets moved."""
    from tinyfaces import transforms
    from tinyfaces.evaluation import _on_device, get_detections
    from tinyfaces.models.model import DetectionModel
    from oracle.targets import RF
    m, _ = models
    g = golden("detections")
    templates = golden("targets")["templates"]
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    m.set_compute_dtype(torch.bfloat16)
    kw = dict(prob_thresh=float(g["thr"]), nms_thresh=0.3, scales=tuple(g["scales"].tolist()), device="cuda", pyramid_on_gpu=True)
    img = torch.from_numpy(g["img"])
    with torch.no_grad(), m.constant_weights(reserve=(1, img.shape[1] * 2, img.shape[2] * 2)):
        a = get_detections(m, img, templates, RF, tf, **kw)
        table, ready = m._table_key, m._ready_key
        assert table is not None and ready is not None
        b = get_detections(m, img, templates, RF, tf, **kw)
        assert m._table_key is table and m._ready_key == ready          # nothing was invalidated, nothing re-packed
    assert np.array_equal(a, b)
    cpu = DetectionModel(num_templates=25)
    assert next(_on_device(cpu, torch.device("cuda")).parameters()).is_cuda and _on_device(m, torch.device("cuda:0")) is m


def test_trainer_two_steps_vs_reference_golden(golden):
    """trainer.train (trainer.py:68-90) with torch.optim.SGD exactly as main.py:67-70 builds it."""
    from tinyfaces import trainer
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel
    from oracle import criterion as ocrit
    g = golden("trainer")
    m = DetectionModel(num_templates=25)
    om = _load_oracle_weights(m)
    m.set_compute_dtype(torch.float32)
    crit = DetectionCriterion(25)
    opt = torch.optim.SGD(m.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    batches = [(torch.from_numpy(g[f"b{i}_img"]), torch.from_numpy(g[f"b{i}_cm"].astype(np.float64)), torch.from_numpy(g[f"b{i}_rm"]))
               for i in range(2)]
    # The reference's balance sampling consumes np.random; replay its draws through the oracle on the
    # oracle model so that the HIP criterion can be fed the identical keep flags.
    class Replay:
        def __init__(self):
            np.random.seed(3)
        def flags(self, out, cm, rm):
            r = ocrit.criterion(out.detach().cpu(), cm.float().cpu().clone(), rm.float().cpu(), want_grad=False)
            B, E = out.shape[0], 25 * out.shape[2] * out.shape[3]
            pk = np.ones((B, E), np.uint8); nk = np.ones((B, E), np.uint8)
            for b, rec in enumerate(r["records"]):
                pk[b, :rec["pos_keep"].size] = rec["pos_keep"]; nk[b, :rec["neg_keep"].size] = rec["neg_keep"]
            return torch.from_numpy(pk), torch.from_numpy(nk)
    rp = Replay()
    orig = crit.forward
    def fwd(out, cm, rm):
        crit.inject_sampling(*rp.flags(out, cm, rm))
        return orig(out, cm, rm)
    crit.forward = fwd
    buf = io.StringIO()
    with redirect_stdout(buf):
        trainer.train(m, crit, opt, batches, 0, torch.device("cuda"))
    lines, ref_lines = buf.getvalue().strip().split("\n"), str(g["log"]).strip().split("\n")
    vals = lambda s: [float(t.split(": ")[1]) for t in s.split("\t") if "loss" in t]
    la, lb = [vals(l) for l in lines], [vals(l) for l in ref_lines]
    sd = m.state_dict()
    worst = 0.0
    for k in g.files:
        if k.startswith("after::"):
            d = err(sd[k.split("::")[1]].cpu().numpy(), g[k])
            worst = max(worst, d[2])
    report("trainer_2steps", log=str(la), ref=str(lb), worst_param_rel=worst)
    assert lines[0].startswith("Epoch: [0][0/2]")
    assert np.allclose(la, lb, rtol=5e-3)
    assert worst < 2e-2      # parameters after two SGD steps; bounded by the ReLU-boundary sensitivity of the gradients (see above)


@pytest.mark.parametrize("stat_rows", [0, 8])
def test_fused_engine_equals_autograd_trainer(golden, stat_rows):
    """TrainEngine (no autograd, flat parameters, fused SGD per group) must take the same step as trainer.train +
    torch.optim.SGD built like main.py:67-70.  With reproducible BN statistics (stat_rows=0: unfolded sums, separate
    finalize kernels) the parameters agree to fp32 rounding (1e-6) after ONE step.  After two steps only to ~1e-3: the
    fp32-atomic summation order of the weight gradients (1e-7) is amplified by batch-stat BN over 128 samples -- the same
    spread is measured between two runs of trainer.train itself (scripts/debug_engine.py).  With the default folded
    statistics (stat_rows=8: fp32 atomics, in-kernel finalize) that amplification already acts on step one."""
    from tinyfaces import _hip, trainer
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel
    g = golden("trainer")
    batches = [(torch.from_numpy(g[f"b{i}_img"]), torch.from_numpy(g[f"b{i}_cm"].astype(np.float32)), torch.from_numpy(g[f"b{i}_rm"]).float())
               for i in range(2)]
    E = 25 * 16 * 16
    keep = torch.ones(2, E, dtype=torch.uint8)          # deterministic sampling: keep the first 128 of each kind
    keep[:, 128:] = 0

    def fresh():
        m = DetectionModel(num_templates=25)
        _load_oracle_weights(m)
        m.set_compute_dtype(torch.float32)
        c = DetectionCriterion(25)
        c.inject_sampling(keep, keep)
        return m, c

    def worst_diff(a, b):
        w, name = 0.0, ""
        for k in a:
            if a[k].is_floating_point():
                d = err(b[k].cpu().numpy(), a[k].cpu().numpy())[2]
                if d > w:
                    w, name = d, k
        return w, name

    res = {}
    prev = _hip.lib().tf_get_stat_rows()
    try:
        _hip.lib().tf_set_stat_rows(stat_rows)
        for nsteps in (1, 2):
            m1, c1 = fresh()
            opt = torch.optim.SGD(m1.learnable_parameters(1e-3), lr=1e-3, momentum=0.9, weight_decay=5e-4)
            with redirect_stdout(io.StringIO()):
                trainer.train(m1, c1, opt, batches[:nsteps], 0, torch.device("cuda"))
            m2, c2 = fresh()
            eng = TrainEngine(m2, c2, lr=1e-3, momentum=0.9, weight_decay=5e-4, device="cuda")
            for img, cm, rm in batches[:nsteps]:
                eng.step(img.cuda(), cm.cuda(), rm.cuda())
            res[nsteps] = worst_diff(m1.state_dict(), m2.state_dict())
            assert int(m2.state_dict()["model.bn1.num_batches_tracked"]) == nsteps
            assert list(m2.state_dict().keys()) == list(m1.state_dict().keys())
    finally:
        _hip.lib().tf_set_stat_rows(prev if prev <= 16 else 0)
    report(f"engine_vs_trainer[rows={stat_rows}]", step1=res[1][0], step1_tensor=res[1][1], step2=res[2][0], step2_tensor=res[2][1])
    assert res[1][0] < (1e-6 if stat_rows == 0 else 2e-3), res[1]
    # step 2 amplifies the step-1 difference by 1e2..1e4 on this 2-image batch (measured 7e-8 -> 1.3e-3 with reproducible statistics,
    # 1e-4 -> 3e-2 with atomically summed ones): the bound only screens for structural errors (wrong group lr, momentum, decay)
    assert res[2][0] < (1e-2 if stat_rows == 0 else 1e-1), res[2]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dual_stream_backward_equals_single_stream(dtype):
    """Race screen: the weight-gradient stream must produce the same gradients as the single-stream order
    (only fp32-atomic summation order may differ), repeated to give a hazard a chance to show."""
    from tinyfaces import _hip
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(dtype).train()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 3, 224, 288, generator=g).cuda()

    def grads(dual):
        m.single_stream = not dual           # (r4: a property of the model's own calls -- tf_detnet_hooks.single_stream -- not a process-wide switch)
        m.zero_grad(set_to_none=True)
        y = m(x)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).cuda()
        y.backward(gy)
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    try:
        _hip.lib().tf_set_stat_rows(0)      # unfolded, plain-store BN statistics + separate finalize kernels: the forward is bit-reproducible
        ref = grads(False)
        worst = 0.0
        for rep in range(4):
            got = grads(True)
            for k in ref:
                d = float((got[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30))
                worst = max(worst, d)
    finally:
        m.single_stream = False
        _hip.lib().tf_set_stat_rows(8)                  # the library default
    report(f"dual_stream[{dtype}]", worst_rel=worst)
    assert worst < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_constant_weights_session_is_bit_identical(models, dtype):
    """Inside model.constant_weights() the packed weights / folded BN of the first forward are reused for every later
    image size (TF_DETNET_WEIGHTS_READY); outputs must equal the re-packing path bit for bit, and the cache must drop
    when the session closes (a weight edit afterwards has to be visible)."""
    m, _ = models
    m.set_compute_dtype(dtype).eval()
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(1, 3, h, w, generator=g).cuda() for h, w in [(96, 128), (200, 168), (64, 64), (200, 168)]]
    with torch.no_grad():
        ref = [m(x).clone() for x in xs]
        with m.constant_weights(reserve=(1, 200, 168)):
            got = [m(x).clone() for x in xs]
            assert m._ready_key is not None
        assert m._ready_key is None
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        # after the session: edits are honoured again
        w = m.score_res3.bias
        old = w.detach().clone()
        try:
            w.data.add_(1.0)
            y2 = m(xs[0])
            assert float((y2 - ref[0]).abs().max()) > 0.5
        finally:
            w.data.copy_(old)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_forward_levels_on_lanes_is_bit_identical(models, dtype):
    """DetectionModel.forward_levels: the pyramid levels of an image side by side on 2-3 HIP streams, each lane with its own workspace
    and packed weights == the sequential loop bit for bit, for every lane count, twice in one session (second pass: WEIGHTS_READY in
    every lane), with a level list whose order is not by size; a weight edit between sessions reaches the lanes' packed copies; outside
    a session / lanes=1 / training mode it IS the sequential loop."""
    m, _ = models
    m.set_compute_dtype(dtype).eval()
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(1, 3, h, w, generator=g).cuda() for h, w in [(64, 96), (260, 200), (128, 136), (33, 47), (200, 260)]]
    try:
        with torch.no_grad():
            ref = [m(x).clone() for x in xs]
            for lanes in (2, 3, 4):
                with m.constant_weights(reserve=(1, 260, 200)):
                    for _ in range(2):
                        got = m.forward_levels(xs, lanes=lanes)
                        torch.cuda.synchronize()
                        for a, b in zip(ref, got):
                            assert torch.equal(a, b), (lanes, float((a - b).abs().max()))
                assert len(m._lanes) >= lanes - 1
            got = m.forward_levels(xs)                     # no session: sequential
            for a, b in zip(ref, got):
                assert torch.equal(a, b)
            w = m.score_res3.bias
            old = w.detach().clone()
            try:
                w.data.add_(1.0)
                with m.constant_weights():
                    y2 = m.forward_levels(xs, lanes=3)
                torch.cuda.synchronize()
                for a, b in zip(ref, y2):
                    assert float((a - b).abs().max()) > 0.5
            finally:
                w.data.copy_(old)
    finally:
        m.set_compute_dtype(torch.float32)


def test_grad_ready_events_are_recorded_in_backward_order():
    """tf_detnet_hooks.events (data-parallel overlap; r4: an argument of the model's own backward call, `model._grad_events`): the executor
    records the caller's events while enqueuing the backward pass.  The event of a LATER bucket (lower block index) must not complete before an earlier one, all of them
    must complete, and once the last (-1) has, the gradients equal those of a run without events."""
    import ctypes as C
    from tinyfaces import _hip
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(torch.bfloat16).train()
    x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(11)).cuda()

    def grads():
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.backward(torch.ones_like(y))
        return y

    blocks_py = [22, 14, 7, -1]
    evs = [torch.cuda.Event(enable_timing=True) for _ in blocks_py]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    blocks = (C.c_int * 4)(*blocks_py)
    handles = (C.c_void_p * 4)(*[int(e.cuda_event) for e in evs])
    try:
        _hip.lib().tf_set_stat_rows(0)          # reproducible BN statistics: two runs differ by weight-gradient atomics only
        m._grad_events = (blocks, handles, 4)
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        grads()
        evs[-1].synchronize()
        got = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        times = [t0.elapsed_time(e) for e in evs]
        m._grad_events = None
        grads()
        torch.cuda.synchronize()
    finally:
        m._grad_events = None
        _hip.lib().tf_set_stat_rows(8)
    assert all(t > 0 for t in times) and times == sorted(times), times
    worst = max(float((got[k] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)) for k, p in m.named_parameters() if p.grad is not None)
    report("grad_events", times_ms=[round(t, 3) for t in times], worst_rel=worst)
    assert worst < 1e-3


def test_context_free_entry_points_equal_the_context_form():
    """include/tinyfaces_hip.h keeps tf_detnet_forward / tf_detnet_backward + the process-wide setters of rounds 1-3 as wrappers over a default
    context: the same gradients as the model's own (context + per-call hooks) path, and a registered event is recorded."""
    import ctypes as C
    from tinyfaces import _hip
    from tinyfaces._hip import lib, ptr, stream
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(torch.bfloat16).train()
    m.flatten_parameters()
    x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(5)).cuda()
    m._sync_tables(x.device)
    try:
        lib().tf_set_stat_rows(0)                # reproducible BN statistics
        out = m._run_forward(x, training=True)
        gout = torch.ones_like(out)
        ref = m._run_backward(x, gout, persistent=True).clone()
        # the same pass through the context-free entry points
        N, _, H, W = x.shape
        bn = m.model.bn1
        out2 = torch.empty_like(out)
        assert lib().tf_detnet_forward(m.compute_dtype, 1, ptr(x), N, H, W, m.num_out, m._param_ptrs, float(bn.eps), float(bn.momentum), ptr(out2),
                                       ptr(m._ws), m._ws.numel(), 0, stream()) == 0
        ev = torch.cuda.Event()
        ev.record(); torch.cuda.synchronize()
        blocks, handles = (C.c_int * 1)(-1), (C.c_void_p * 1)(int(ev.cuda_event))
        g2 = torch.zeros_like(ref)
        table = (C.c_void_p * len(m._names))(*[(g2.data_ptr() + 4 * m._segments[k][0]) if k in m._segments else 0 for k in m._names])
        assert lib().tf_detnet_set_grad_events(blocks, handles, 1) == 0
        try:
            assert lib().tf_detnet_backward(m.compute_dtype, ptr(x), N, H, W, m.num_out, m._param_ptrs, table, ptr(gout), ptr(g2), g2.numel() * 4,
                                            ptr(m._ws), m._ws.numel(), stream()) == 0
        finally:
            lib().tf_detnet_set_grad_events(None, None, 0)
        ev.synchronize()                          # the registered event was recorded behind the whole pass
        torch.cuda.synchronize()
    finally:
        lib().tf_set_stat_rows(8)
    assert torch.equal(out2, out)
    rel = float((g2 - ref).abs().max() / ref.abs().max())
    report("context_free_entry_points", rel=rel)
    assert rel < 1e-3                             # (fp32 atomics of the non-grouped weight gradients)


def test_grad_ready_callback_is_called_per_bucket_with_the_carrying_stream():
    """tf_detnet_hooks.fn (r3: tf_detnet_set_grad_callback): while it enqueues the backward pass the executor calls the registered function once per registered
    block, in backward order, with the stream that carries that bucket's gradients -- its second stream for the bottleneck buckets, the
    caller's stream for the final one (-1) -- and with NULL events (callback-only registration).  Work enqueued by the callee on that stream
    (here: a snapshot copy of a bucket's gradient slice) sees the bucket final; an exception inside the callback does not unwind through
    the C frames (the engine stores it: tinyfaces/engine.py:_on_bucket); without registered blocks the callback is never called."""
    import ctypes as C
    from tinyfaces import _hip
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(torch.bfloat16).train()
    x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(12)).cuda()
    m.flatten_parameters()
    m._sync_tables(x.device)
    out = m._run_forward(x, training=True)
    blocks_py = [22, 14, 7, -1]
    blocks = (C.c_int * 4)(*blocks_py)
    nulls = (C.c_void_p * 4)(None, None, None, None)
    calls, snaps = [], {}
    main = torch.cuda.current_stream().cuda_stream
    seg = m._segments
    o22 = min(o for k, (o, _) in seg.items() if k.startswith("model.layer3.22."))

    def cb(block, stream, _user):
        calls.append((block, stream or 0))
        if block == 22:                                                   # everything from layer3.22 on (incl. the heads) is final on `stream`
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                snaps[22] = m._grad_flat_persistent[o22:].clone()

    fn = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_void_p)(cb)
    m._grad_events = (blocks, nulls, 4)
    m._grad_callback = fn
    try:
        g = m._run_backward(x, torch.ones_like(out), persistent=True)
        torch.cuda.synchronize()
    finally:
        m._grad_events = None
        m._grad_callback = None
    assert [b for b, _ in calls] == blocks_py, calls
    assert calls[-1][1] == main and all(s != main for _, s in calls[:-1]), (calls, main)
    assert len({s for _, s in calls[:-1]}) == 1                           # one second stream
    assert torch.equal(snaps[22], g[o22:])                                # the bucket was final where the callback was told it is
    # no registration -> no call
    calls.clear()
    m._run_forward(x, training=True)
    m._run_backward(x, torch.ones_like(out), persistent=True)
    torch.cuda.synchronize()
    assert calls == []


_REFUSED_GROUP_SCRIPT = r'''
import sys, numpy as np, torch
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd"]
from oracle.model import OracleDetectionModel, tame_init_
from tinyfaces.engine import TrainEngine
from tinyfaces.models.loss import DetectionCriterion
from tinyfaces.models.model import DetectionModel
g = np.load(ROOT + "/tests/golden/trainer.npz")
img, cm, rm = (torch.from_numpy(g["b0_img"]).cuda(), torch.from_numpy(g["b0_cm"].astype(np.float32)).cuda(), torch.from_numpy(g["b0_rm"]).float().cuda())
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict())
m.set_compute_dtype(torch.bfloat16)
c = DetectionCriterion(25)
keep = torch.ones(2, 25 * 16 * 16, dtype=torch.uint8); keep[:, 128:] = 0
c.inject_sampling(keep, keep)
eng = TrainEngine(m, c, lr=0.0, momentum=0.0, weight_decay=0.0, device="cuda")     # lr 0: both steps differentiate the SAME function
out = []
for _ in range(2):
    eng.step(img, cm.clone(), rm)
    torch.cuda.synchronize()
    out.append(m._grad_flat_persistent.detach().float().cpu().numpy().copy())
o, n = m._segments["model.layer3.5.conv2.weight"]
np.savez(sys.argv[1], g1=out[0], g2=out[1], lo=o, n=n)
'''


@pytest.mark.parametrize("refuse", [0, 1])
def test_refused_wgrad_group_falls_back_into_zeroed_gradients(tmp_path, refuse):
    """when tf_conv2d_wgrad_group refuses a group (layer-3 width the all-taps plan does not take), the per-problem
    fallback accumulates with fp32 atomics -- into tensors the split flat-gradient memset skipped.  On the engine's PERSISTENT gradient the
    second step would then carry step one's values on top.  TINYFACES_DBG_GROUP_REFUSE=1 forces the refusal at any size (the knob is read
    once per process, hence the subprocess): with lr = 0 the two steps' gradients must agree, and must equal the grouped kernels'."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"g{refuse}.npz")
    env = dict(os.environ)
    env.pop("TINYFACES_DBG_GROUP_REFUSE", None)
    if refuse:
        env["TINYFACES_DBG_GROUP_REFUSE"] = "1"
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _REFUSED_GROUP_SCRIPT, out], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    g1, g2, lo, n = z["g1"], z["g2"], int(z["lo"]), int(z["n"])
    a, b = g1[lo:lo + n], g2[lo:lo + n]
    assert np.abs(a).max() > 0
    cos = float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))
    ratio = float(np.linalg.norm(b) / np.linalg.norm(a))
    cos_all = float(np.dot(g1, g2) / (np.linalg.norm(g1) * np.linalg.norm(g2)))
    report(f"refused_group[refuse={refuse}]", cos_layer3_conv2=cos, norm_ratio=ratio, cos_flat=cos_all)
    # the same function twice.  Not bit-equal: the fp32 atomics of the BN statistic rows land in another order, and on this 2-image
    # fixture (batch statistics over <= 512 pixels) that flips bf16 roundings all the way up -- a layer-3 3x3 gradient repeats itself to
    # cosine 0.978 with or without the refusal.  A gradient accumulated onto the previous step's would have norm ratio 2.
    assert 0.97 < ratio < 1.03 and cos > 0.95, (ratio, cos)
    assert cos_all > 0.999


def test_eval_pyramid_is_bit_reproducible(templates):
    """r5: `get_detections` on ONE image, thirty times, each call with its own constant-weights session and its three pyramid levels side by
    side on the model's lanes -- the candidate list (every box and score of every level, before NMS) and the kept indices must be the same
    arrays every time.  Before the conv3x3h fix (csrc/conv3x3h.hip wait_vmcnt: reads of a ring slot in flight across the barrier that
    frees it) about one run in four differed in the last few candidates of the largest level: the 3x3 convolution of a layer-3 bottleneck
    returned a few different values when three forwards shared the GPU (scripts/diag_arena_diff.py finds the first differing tensor)."""
    from oracle.targets import RF
    from tinyfaces import transforms
    from tinyfaces.evaluation import get_detections
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    g = torch.Generator().manual_seed(4)
    img = torch.rand(3, 700, 933, generator=g)
    with torch.no_grad():
        y = m(torch.randn(1, 3, 700, 933, generator=g).cuda()).cpu()
    thr = float(torch.quantile(torch.sigmoid(y[0, :25]).flatten()[:2000000], 0.999))
    kw = dict(prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device="cuda", pyramid_on_gpu=True, return_candidates=True)
    r0, c0, k0 = get_detections(m, img, templates, RF, tf, **kw)
    assert c0.shape[0] > 500
    differing = 0
    for _ in range(30):
        r, c, k = get_detections(m, img, templates, RF, tf, **kw)
        differing += int(c.shape != c0.shape or not np.array_equal(c, c0) or not np.array_equal(k, k0))
    report("eval_pyramid_reproducible", candidates=int(c0.shape[0]), kept=int(k0.shape[0]), differing_runs=differing)
    assert differing == 0


@pytest.mark.parametrize("training", [False, True])
def test_forward_is_bit_identical_beside_resource_hogs(training):
    """r5, after the conv3x3h race: the detector's forward must return the SAME bits whatever else the GPU is doing.  The pass runs alone
    (reference), then fifteen times while two more streams keep the CUs busy with the single-resource probes of csrc/probe.hip -- LDS-DMA
    traffic with 96 KiB of LDS per block, LDS reads, HBM streams, MFMA chains, parked waves holding LDS -- i.e. with its blocks sharing
    CUs, LDS, the LDS-DMA path and the memory system with foreign kernels at unpredictable phases.  Eval mode (folded BN) and training
    mode with bit-reproducible statistic rows (tf_set_stat_rows(0): no atomics in the forward).  A general guard against kernels that
    only order their LDS traffic by luck.  (Measured with the UNFIXED conv3x3h: these probes did not trigger its race in 15 runs -- two
    more detector forwards on other streams do, 6 runs of 30: test_eval_pyramid_is_bit_reproducible is the test with teeth for that one.)"""
    from tinyfaces import _hip
    from tinyfaces.models.model import DetectionModel
    lib = _hip.lib()
    m = DetectionModel(num_templates=25)
    _load_oracle_weights(m)
    m = m.cuda().set_compute_dtype(torch.bfloat16)
    m = m.train() if training else m.eval()
    m.model.bn1.momentum = 0.0                             # (training mode: the running means are the shift of the statistic sums -- keep them fixed between runs)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3 if training else 1, 3, 500 if training else 1000, 500 if training else 1333, generator=g).cuda()
    hogs = [torch.cuda.Stream(), torch.cuda.Stream()]
    win = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    prev = lib.tf_get_stat_rows()

    def run():
        with torch.no_grad():
            return m._run_forward(x, training=training).clone() if training else m(x).clone()

    def hog(round_):
        kinds = [(4, 256, 96 * 1024, 40), (6, 256, 64 * 1024, 400), (2, 512, 0, 60), (3, 256, 0, 3000), (0, 256, 120 * 1024, 200)]
        for si, s in enumerate(hogs):
            kind, blocks, lds, iters = kinds[(round_ + 2 * si) % len(kinds)]
            _hip.check(lib.tf_debug_probe_chain(kind, blocks, lds, win.data_ptr(), win.numel(), iters, 12, s.cuda_stream), "probe")
    try:
        if training:
            lib.tf_set_stat_rows(0)
            m._sync_tables(x.device)
        ref = run()
        torch.cuda.synchronize()
        assert torch.equal(run(), ref)                     # (alone it repeats itself)
        bad = []
        for r in range(15):
            hog(r)
            y = run()
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad.append((r, float((y - ref).abs().max())))
    finally:
        lib.tf_set_stat_rows(prev if prev <= 16 else 0)
    report(f"forward_beside_hogs[training={training}]", differing_runs=len(bad), worst=max([b[1] for b in bad], default=0.0))
    assert not bad, bad


_FUSED_PROLOGUE_SCRIPT = r'''
import sys, numpy as np, torch
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd"]
from oracle.model import OracleDetectionModel, tame_init_
from tinyfaces.models.model import DetectionModel
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict())
m = m.cuda().set_compute_dtype(torch.bfloat16).train()
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 3, 160, 224, generator=g).cuda()
y = m(x)
gy = (torch.randn(y.shape, generator=g) * 0.1).cuda()
y.backward(gy)
names = ["model.conv1.weight", "model.layer2.2.conv1.weight", "model.layer3.4.conv1.weight", "model.layer3.9.conv3.weight", "model.layer3.20.bn3.weight", "score_res4.weight"]
pd = dict(m.named_parameters())
np.savez(sys.argv[1], y=y.detach().float().cpu().numpy(), rm=m.model.layer3[6].bn3.running_mean.detach().cpu().numpy(),
         **{f"g{i}": pd[k].grad.float().cpu().numpy() for i, k in enumerate(names)})
'''


def test_opt_in_fused_bn_prologues_take_the_same_step(tmp_path):
    """r5: the executor with the fused BatchNorm prologues switched ON (TINYFACES_PWX_FWD=1: bn3 + residual + ReLU of a bottleneck on the
    operand path of the next conv1, tf_conv2d_bnfwd; TINYFACES_PWX_ALL=1: the BN-backward apply on the operand path of conv3's data gradient,
    tf_conv2d_bnbwd) against the default graph (every pass a launch of its own), one training forward + backward from the same weights: the
    maps, the running statistics the fused kernel publishes, and gradients of layers 1-3 + heads.  The knobs are read once per process: two
    subprocesses.  (The fused graph is opt-in because it is slower; it has to stay CORRECT.)"""
    from tinyfaces import _hip as _hip_x
    if not _hip_x.experimental():
        pytest.skip("the fused BatchNorm prologues (conv_pwx) are compiled into the experimental build only (build.py --experimental)")
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, extra in (("default", {}), ("fused", {"TINYFACES_PWX_FWD": "1", "TINYFACES_PWX_ALL": "1"})):
        env = {k: v for k, v in os.environ.items() if not k.startswith("TINYFACES_PWX")}
        env.update(extra)
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _FUSED_PROLOGUE_SCRIPT, out], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[tag] = np.load(out)
    a, b = outs["default"], outs["fused"]
    span = float(np.abs(a["y"]).max())
    d_y = float(np.abs(a["y"] - b["y"]).max()) / span
    d_rm = float(np.abs(a["rm"] - b["rm"]).max() / (np.abs(a["rm"]).max() + 1e-30))
    cos = []
    for i in range(6):
        u, v = a[f"g{i}"].ravel().astype(np.float64), b[f"g{i}"].ravel().astype(np.float64)
        cos.append(float(u @ v / (np.linalg.norm(u) * np.linalg.norm(v) + 1e-300)))
    report("fused_bn_prologues_vs_default", maps_rel=d_y, running_mean_rel=d_rm, grad_cos_min=min(cos), grad_cos=[round(c, 4) for c in cos])
    assert d_y < 2e-2 and d_rm < 1e-3                     # bf16 maps of two summation orders; the published statistics are fp32
    assert min(cos) > 0.95, cos                           # measured 0.974 (two runs of the DEFAULT graph agree to 0.978 on this fixture: fp32 atomics + bf16)
