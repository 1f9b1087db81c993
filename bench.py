#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:

    train img/s @ 500x500, bs=12/GPU  (configs[2]: synthetic crops + random boxes, dense_overlap targets on
    the GPU, forward, criterion, backward, [RCCL gradient all-reduce], fused SGD), bf16 MFMA compute.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `value` = whole-job images/s with every input already resident in HBM when the timed
region starts.  Extra objects: `roofline` (dominant MFMA kernel, HIP-event timed inside the timed region),
`cpu_baseline` (the CPU oracle = port of the reference's torch-CPU path, bounded sample, N=1 only) and `eval`
(configs[1]: ms/image for a 3-scale pyramid of a 1280x960 image: forward x3 + decode + NMS).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}       # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_IMG = 73.559                           # BASELINE.md section 3 (2*MACs of all 97 convs @500x500)
PROFILE_EVERY = 11                                    # roofline timing: HIP events around every 11th MFMA launch (287 launches/step is not a multiple: the sample rotates over the layers)
KIND_NAMES = {0: "conv_igemm<f32,128x128>", 1: "conv_igemm<f32,128x64>", 2: "conv_igemm<f32,64x64>",
              3: "conv_igemm<bf16,128x128>", 4: "conv_igemm<bf16,128x64>", 5: "conv_igemm<bf16,64x64>",
              8: "wgrad<f32,64>", 9: "wgrad<f32,128>", 10: "wgrad<bf16,64>", 11: "wgrad<bf16,128>",
              12: "conv_dma<f32,64x64>", 13: "conv_dma<bf16,64x64>", 14: "wgrad_dma<bf16,64x64x3>"}      # conv_dma: ring depth 1/2/3 by K (csrc/conv_dma.hip)


def tame_init_(model, seed=0):
    """Random-init weights of the real architecture (no checkpoint here).  Plain kaiming init saturates the sigmoid
    (SURVEY.md section 7.1), so bn3.weight = 0.1 and the head weights are scaled by 0.05."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bn3.weight"):
                p.fill_(0.1)
        for name in ("score_res3", "score_res4"):
            getattr(model, name).weight.mul_(0.05)
    return model


def synthetic_batch(seed, bs, device, templates_d):
    """randn images; per image G~U{1..16} boxes, width log-uniform 8-200 px, h/w~U[1,1.5]."""
    from tinyfaces.datasets.synthetic import random_boxes
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(bs, 3, 500, 500, generator=g).to(device)
    rng = np.random.RandomState(seed)
    boxes = [random_boxes(rng) for _ in range(bs)]
    offs = np.cumsum([0] + [b.shape[0] for b in boxes]).astype(np.int32)
    return dict(x=x, boxes=torch.from_numpy(np.concatenate(boxes)).to(device), offs=torch.from_numpy(offs).to(device), total=int(offs[-1]),
                paste=torch.tensor([[0, 0, 500, 500]] * bs, dtype=torch.int32, device=device), host_boxes=boxes)


def cpu_baseline(bs=1, steps=1):
    """The CPU oracle (restatement of the reference's torch-CPU path, validated against the reference's golden
    vectors) timed on this host: target assignment (vectorised numpy) + forward + criterion + backward + SGD."""
    from oracle import criterion as ocrit
    from oracle import targets as otgt
    from oracle.model import OracleDetectionModel, tame_init_ as o_tame
    from tinyfaces.datasets.synthetic import random_boxes
    from tinyfaces.datasets.templates import load_templates
    threads = min(32, os.cpu_count() or 1)       # torch-CPU convs stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(threads)
    t = load_templates()
    m = o_tame(OracleDetectionModel(num_templates=25), 0).train()
    opt = torch.optim.SGD(m.learnable_parameters(1e-4), lr=1e-4, momentum=0.9, weight_decay=5e-4)
    pad = otgt.get_padding(t, [0, 0, 500, 500])
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)
    times = []
    for it in range(steps + 1):
        x = torch.randn(bs, 3, 500, 500, generator=g)
        boxes = [random_boxes(rng) for _ in range(bs)]
        t0 = time.perf_counter()
        maps = [otgt.get_heatmaps(b.copy(), t, pad) for b in boxes]
        cm = torch.from_numpy(np.ascontiguousarray(np.stack([c.transpose(2, 0, 1) for c, _, _ in maps]))).float()
        rm = torch.from_numpy(np.ascontiguousarray(np.stack([r.transpose(2, 0, 1) for _, r, _ in maps]))).float()
        out = m(x)
        r = ocrit.criterion(out, cm, rm)
        opt.zero_grad()
        out.backward(r["grad"])
        opt.step()
        times.append(time.perf_counter() - t0)
        if times[0] > 25.0:            # bounded sample: on a slow host the (cold) first step is the measurement
            break
    dt = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {"value": round(bs / dt, 3), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"{steps} timed steps (1 warm-up) of bs={bs} 500x500: numpy target assignment + torch-CPU fp32 fwd/loss/bwd/SGD"}


PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01e_pmc_traffic.json")
PMC_PATTERNS = {13: "conv_dma_kernel<tf::bf16_t, 64, 64,", 12: "conv_dma_kernel<float, 64, 64,", 14: "wgrad_dma_kernel"}


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (scripts/gpu_pmc_bench.sh: FETCH_SIZE x2 + WRITE_SIZE, separate passes).  Counters cannot be read from inside the
    process, so this is the offline measurement, labelled as such; None when the file or the kernel is missing."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            data = json.load(f)
        pat = PMC_PATTERNS[kind]
        rows = [v for k, v in data["kernels"].items() if k.startswith(pat)]
        n = sum(r["launches"] for r in rows)
        return round(sum(r["hbm_bytes"] * r["launches"] for r in rows) / n) if n else None
    except Exception:
        return None


EPI_NAMES = [(1, "affine"), (2, "res"), (4, "relu"), (8, "stats"), (16, "mask"), (32, "stats2"), (64, "join"), (128, "mask2"), (256, "stats3")]


def write_layer_table(_hip, path, steps, dt):
    """Per layer shape: launches per step, average launch duration (HIP events around every launch), achieved TFLOP/s and GB/s
    (algorithmic), and the two roofline times of that shape: flops / 2.5 PFLOP/s and bytes / 6.3 TB/s (achievable HBM)."""
    rows = (C.c_double * (11 * 256))()
    n = _hip.lib().tf_profile_shapes(rows, 256)
    out = []
    for i in range(n):
        kind, M, N, K, taps, mode, epi, launches, ms, flops, nbytes = [rows[i * 11 + j] for j in range(11)]
        us = ms * 1e3 / launches
        f1, b1 = flops / launches, nbytes / launches
        t_mfma, t_hbm = f1 / 2.5e15 * 1e6, b1 / 6.3e12 * 1e6
        out.append({"kernel": KIND_NAMES.get(int(kind), str(int(kind))), "op": ["conv", "dgrad", "wgrad"][int(mode)], "M": int(M), "N": int(N), "K": int(K),
                    "taps": int(taps), "epilogue": "+".join(nm for bit, nm in EPI_NAMES if int(epi) & bit) or "-",
                    "launches_per_step": round(launches / steps, 2), "avg_us": round(us, 2), "ms_per_step": round(ms / steps, 3),
                    "tflops": round(f1 / us / 1e6, 1), "gb_s": round(b1 / us / 1e3, 1), "mfma_bound_us": round(t_mfma, 2), "hbm_bound_us": round(t_hbm, 2),
                    "x_over_roofline": round(us / max(t_mfma, t_hbm), 1)})
    out.sort(key=lambda r: -r["ms_per_step"])
    with open(path, "w") as f:
        json.dump({"note": "HIP events around every launch (perturbs the step by ~10 %); roofline = max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s)",
                   "ms_per_step_with_events": round(dt / steps * 1e3, 3), "shapes": out}, f, indent=1)
    md = os.path.splitext(path)[0] + ".md"
    with open(md, "w") as f:
        f.write("| op | M | N | K | taps | epilogue | launches/step | avg us | ms/step | TFLOP/s | GB/s | MFMA-bound us | HBM-bound us | x over roofline |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in out:
            f.write("| {op} | {M} | {N} | {K} | {taps} | {epilogue} | {launches_per_step} | {avg_us} | {ms_per_step} | {tflops} | {gb_s} | {mfma_bound_us} | "
                    "{hbm_bound_us} | {x_over_roofline} |\n".format(**r))


def bench_eval(model, templates, device, runs=20):       # warm, median of >= 20
    """configs[1]: 1280x960 image, 3-scale pyramid (480x640, 960x1280, 1920x2560): forward x3 + decode + one NMS."""
    from tinyfaces import ops
    model.eval()
    g = torch.Generator().manual_seed(0)
    levels = [(0.5, torch.randn(1, 3, 480, 640, generator=g).to(device)), (1, torch.randn(1, 3, 960, 1280, generator=g).to(device)),
              (2, torch.randn(1, 3, 1920, 2560, generator=g).to(device))]
    t_d = torch.as_tensor(templates, dtype=torch.float64).to(device)
    cap = sum((x.shape[2] // 8 + 1) * (x.shape[3] // 8 + 1) for _, x in levels) * 25
    dets = torch.empty(cap, 5, dtype=torch.float64, device=device)
    masks = {s: [torch.from_numpy(a).to(device) for a in ops.template_masks(templates, s, (x.shape[3] + 7) // 8, "w")] for s, x in levels}
    thr = None
    times, n_cand, n_keep = [], 0, 0
    with torch.no_grad(), model.constant_weights(reserve=(1, 1920, 2560)):       # what evaluate_model.py's image loop does
        for it in range(runs + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            count = torch.zeros(1, dtype=torch.int32, device=device)
            outs = []
            for s, x in levels:
                out = model(x)
                if thr is None:
                    outs.append(out)
                    continue
                ops.decode_compact(out[0], t_d, masks[s][0], masks[s][1], thr, s, dets, count)
            if thr is None:        # calibrate the threshold once so that N is a few thousand candidates (random weights)
                allp = torch.cat([torch.sigmoid(o[0, :25]).flatten() for o in outs])
                thr = float(torch.quantile(allp[torch.randperm(allp.numel(), device=device)[:1000000]], 0.995))
                continue
            n = int(count.item())
            keep = ops.nms(dets[:n, :4].contiguous(), dets[:n, 4].contiguous(), 0.3)
            res = dets[:n][keep].cpu()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            n_cand, n_keep = n, res.shape[0]
    ms = float(np.median(times[1:])) * 1e3
    gflop = 1829.4
    return {"ms_per_image": round(ms, 3), "pyramid": "480x640+960x1280+1920x2560", "candidates": n_cand, "kept": n_keep,
            "achieved_tflops": round(gflop / ms, 2), "frac_of_bf16_mfma_peak": round(gflop / ms / PEAK_TFLOPS["bf16"], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)        # >= 50 steps after 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--with-augment", action="store_true",
                    help="SURVEY 8f.1 in the loop: every step also builds its batch from resident 768x1024 uint8 images "
                         "(resize x0.5/x1/x2, crop, paste, flip, normalise on the GPU + box bookkeeping on the host)")
    ap.add_argument("--layer-table", default=None, metavar="PATH",
                    help="bracket EVERY MFMA launch with HIP events (slows the step by ~10 %%) and write the per-layer-shape roofline "
                         "table (JSON + markdown next to it) instead of sampling 1 launch in 11")
    ap.add_argument("--eval-only", action="store_true", help="only the configs[1] pyramid leg (for rocprofv3 runs of the eval path)")
    args = ap.parse_args()

    from tinyfaces import _hip, ops, parallel
    from tinyfaces.datasets.templates import load_templates
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel

    # stdout carries exactly ONE JSON line: libraries (RCCL prints a version banner through C stdio) get stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("TINYFACES_BENCH_SHARE_GPU") == "1"      # functional test of the N>1 path on a 1-GPU box (gloo, shared device)
    if world > 1:
        parallel.init_from_env("gloo" if share else "nccl")
    elif os.environ.get("TINYFACES_FORCE_DIST") == "1":      # test knob: the RCCL + event-overlap path with a 1-rank group on one GPU
        parallel.init_from_env("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device(f"cuda:{local % torch.cuda.device_count() if share else local}")
    torch.cuda.set_device(device)
    _hip.lib()                                           # fail loudly if the HIP library is missing
    if os.environ.get("TINYFACES_STAT_ROWS"):            # tuning knob: BN statistic partial rows (default 8)
        _hip.lib().tf_set_stat_rows(int(os.environ["TINYFACES_STAT_ROWS"]))

    templates = load_templates()
    t_d = torch.as_tensor(templates, dtype=torch.float64).to(device)
    torch.manual_seed(0)
    model = tame_init_(DetectionModel(num_objects=1, num_templates=25)).set_compute_dtype(args.dtype)
    if args.eval_only:
        os.write(json_fd, (json.dumps({"eval": bench_eval(model.to(device), templates, device, runs=10)}) + "\n").encode())
        return
    crit = DetectionCriterion(25, seed=rank, lazy_meters=True)
    eng = TrainEngine(model, crit, lr=1e-4, momentum=0.9, weight_decay=5e-4, device=device)

    pool = [synthetic_batch(1000 * s + rank, args.batch, device, t_d) for s in range(4)]

    raw = None
    if args.with_augment:
        from tinyfaces.datasets import augment as aug
        from tinyfaces.datasets.synthetic import random_boxes
        gi = torch.Generator().manual_seed(7 + rank)
        raw = [(torch.randint(0, 256, (768, 1024, 3), dtype=torch.uint8, generator=gi).to(device),
                random_boxes(np.random.RandomState(50 + k + 100 * rank)) * np.array([1024 / 500, 768 / 500] * 2)) for k in range(args.batch)]
        arng = np.random.RandomState(rank)

    in_stream = torch.cuda.Stream(device=device) if raw is not None else None

    def build_batch(i):
        """Input pipeline of one step on its OWN stream: its small blocking H2D copies (boxes, offsets) then wait for this
        stream only, not for the training stream that is still busy with the previous step."""
        with torch.cuda.stream(in_stream):
            x = torch.empty(args.batch, 3, 500, 500, device=device)
            boxes, pastes, flips = [], [], []
            for k, (u8, bb) in enumerate(raw):
                _, b2, paste, flip = aug.process_inputs(u8, bb, rng=arng, out=x[k])
                boxes.append(b2); pastes.append(paste); flips.append(int(flip))
            cm, rm = ops.dense_overlap_targets(boxes, templates, paste_boxes=pastes, flips=flips, seed=i * world + rank, device=device)
        return x, cm, rm

    def step(i):
        if raw is not None:               # the real input pipeline of a training step, minus the JPEG decode
            x, cm, rm = build_batch(i)
            cur = torch.cuda.current_stream()
            cur.wait_stream(in_stream)
            for t in (x, cm, rm):
                t.record_stream(cur)
            return eng.step(x, cm, rm)
        b = pool[i % len(pool)]
        cm, rm = ops.dense_overlap_targets_device(b["boxes"], b["offs"], b["total"], t_d, paste_d=b["paste"], seed=i * world + rank)
        return eng.step(b["x"], cm, rm)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if not args.no_profile:
        _hip.lib().tf_profile_enable(1 if args.layer_table else PROFILE_EVERY)     # HIP events around 1 MFMA launch in PROFILE_EVERY (an event pair costs ~5 us of stream time)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss2 = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _hip.lib().tf_profile_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_v = loss2.tolist()

    # ---- N > 1: size of the exchange (achieved bus GB/s and the fraction hidden under the backward pass).  Outside
    #      the timed region, on every rank (collectives), and never allowed to cost the headline line
    comm = None
    if world > 1:
        try:
            dist = torch.distributed
            g = eng.model._grad_flat_persistent
            for _ in range(2):
                dist.all_reduce(g)
            torch.cuda.synchronize(); dist.barrier()
            tc = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(g)
            torch.cuda.synchronize()
            t_ar = (time.perf_counter() - tc) / 5
            eng.skip_allreduce = True
            for i in range(2):
                step(1000 + i)
            torch.cuda.synchronize(); dist.barrier()
            tc = time.perf_counter()
            for i in range(8):
                step(1002 + i)
            torch.cuda.synchronize()
            t_nc = (time.perf_counter() - tc) / 8
            eng.skip_allreduce = False
            both = torch.tensor([t_ar, t_nc], dtype=torch.float64, device=device)
            dist.all_reduce(both, op=dist.ReduceOp.MAX)
            t_ar, t_nc = [float(v) for v in both.tolist()]
            nbytes = g.numel() * 4
            exposed = max(0.0, dt / args.steps - t_nc)
            comm = {"gradient_bytes": nbytes, "allreduce_ms_standalone": round(t_ar * 1e3, 3),
                    "bus_gb_s": round(2 * (world - 1) / world * nbytes / t_ar / 1e9, 1),
                    "ms_per_step_without_exchange": round(t_nc * 1e3, 3), "exposed_ms_per_step": round(exposed * 1e3, 3),
                    "hidden_fraction": round(max(0.0, 1.0 - exposed / t_ar), 3) if t_ar > 0 else None,
                    "buckets": len(eng._overlap["ranges"]) if eng._overlap else None}
        except Exception as e:      # the measurement is a bonus: report the failure, keep the headline
            comm = {"error": repr(e)}

    rows = (C.c_double * (16 * 5))()
    n = _hip.lib().tf_profile_collect(rows, 16)
    prof = [dict(kind=int(rows[i * 5]), launches=int(rows[i * 5 + 1]), ms=rows[i * 5 + 2], flops=rows[i * 5 + 3], bytes=rows[i * 5 + 4])
            for i in range(n)]
    if rank != 0:
        return
    if args.layer_table:
        write_layer_table(_hip, args.layer_table, args.steps, dt)
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    out = {"metric": "train img/s @ 500x500 bs=12/GPU", "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "configs[2]: bs=12/GPU synthetic 500x500 crops + random boxes; dense_overlap targets on GPU, "
                                  "ResNet-101 hybrid-head fwd, criterion, bwd, fused SGD" + (", RCCL grad all-reduce" if world > 1 else "") +
                                  (" + GPU augmentation of 768x1024 uint8 images (8f.1) every step" if args.with_augment else ""),
                      "global_batch": args.batch * world, "image": "500x500", "templates": 25, "parallelism": f"dp{world}",
                      "weights": "random init (tamed kaiming), fp32 master + " + args.dtype + " MFMA operands"},
           "loss": {"cls": round(loss_v[0], 3), "reg": round(loss_v[1], 3)},
           "step_tflops": round(3 * FWD_GFLOP_PER_IMG * args.batch / ms_per_step, 2)}
    if prof:
        dom = max(prof, key=lambda r: r["ms"])
        peak = PEAK_TFLOPS["bf16" if dom["kind"] in (3, 4, 5, 10, 11, 13, 14) else "fp32"]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": KIND_NAMES.get(dom["kind"], str(dom["kind"])), "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                           "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom["kind"]),
                           "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this command (profiles/r01e_pmc_traffic.json); "
                                           "algorithmic bytes per launch = %d" % round(dom["bytes"] / dom["launches"]),
                           "launches": dom["launches"],
                           "sampling": f"HIP events around 1 launch in {PROFILE_EVERY} over the timed region",
                           "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                           "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
                           "algorithmic_gb_s": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                           "share_of_timed_region": round(dom["ms"] * PROFILE_EVERY / (dt * 1e3), 3)}
        out["kernels"] = [{"kernel": KIND_NAMES.get(r["kind"], str(r["kind"])), "launches": r["launches"], "ms_per_step": round(r["ms"] * PROFILE_EVERY / args.steps, 3),
                           "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1), "gb_s": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in prof]
    if comm is not None:
        out["allreduce"] = comm
    if world == 1 and not args.no_eval:
        try:
            out["eval"] = bench_eval(model, templates, device)
        except Exception as e:   # the headline number must still be printed
            out["eval"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
