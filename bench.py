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
PROFILE_EVERY = 23                                    # roofline timing: HIP events around every 23rd MFMA launch (287 launches per step is not a multiple: the sample rotates over the layers; 11 in rounds 1-2, ~1 % of the step)
# kernel kinds of tf_profile_collect (csrc/profile.hip): the executor only launches 12-15; 0-11 are the register-staged kernels kept for the C ABI
KIND_NAMES = {20: "stem_conv<bf16>", 21: "stem_conv<f16>", 22: "stem_wgrad<bf16>", 6: "conv3x3h<bf16>", 7: "conv3x3h<f16>", 12: "conv_dma<f32>", 13: "conv_dma<bf16>", 14: "wgrad_dma<bf16>", 15: "conv_dma<f16>", 16: "wgrad3x3<bf16>", 17: "conv_pwx<bf16>", 18: "wgrad_group<bf16>", 19: "wgrad3x3_group<bf16>", 23: "conv_pws<bf16|f16>",
              8: "wgrad<f32,64>", 9: "wgrad<f32,128>", 10: "wgrad<bf16,64>", 11: "wgrad<bf16,128>"}
BF16_KINDS = (3, 4, 5, 6, 7, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23)


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


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(bs=12, warmup=2, steps=5, eval_warmup=1, eval_runs=5, threads=None):
    """The CPU oracle (restatement of the reference's torch-CPU path, validated against the reference's golden vectors) timed on
    this host, BASELINE.md section 4: (i) a bs=12 training step on synthetic 500x500 crops: target assignment (vectorised numpy)
    + forward + criterion + backward + SGD; (ii) get_detections end to end (PIL pyramid, three forwards, decode, CPU NMS) on a
    seeded 1280x960 image with scales (-1, 0, 1).  Default sample: 2 + 5 training steps (BASELINE.md section 4 asks 3 + 5) and
    1 + 5 images (3 + 20), medians of the timed ones, ~2.5 minutes of host time on the 2 x EPYC 9575F of the GPU boxes and capped
    by TINYFACES_CPU_BUDGET_S (default 200 s, the sample then says how many ran); --cpu-full runs the full 3 + 5 / 3 + 20 protocol."""
    from oracle import criterion as ocrit
    from oracle import pyramid as opyr
    from oracle import targets as otgt
    from oracle.model import OracleDetectionModel, tame_init_ as o_tame
    from oracle.refstub import Compose, Normalize, ToTensor
    from tinyfaces.datasets.synthetic import random_boxes
    from tinyfaces.datasets.templates import load_templates
    # torch-CPU convolutions stop scaling well below the 100+ hardware threads of the GPU hosts and THRASH when given all of them
    # (the bs=12 step did not finish in 10 minutes with os.cpu_count() threads on a 2-socket box): 64 by default, measured best
    threads = threads or int(os.environ.get("TINYFACES_CPU_THREADS", min(64, os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    t = load_templates()
    m = o_tame(OracleDetectionModel(num_templates=25), 0).train()
    opt = torch.optim.SGD(m.learnable_parameters(1e-4), lr=1e-4, momentum=0.9, weight_decay=5e-4)
    pad = otgt.get_padding(t, [0, 0, 500, 500])
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)
    times, t_tgt = [], []
    budget, t_start = float(os.environ.get("TINYFACES_CPU_BUDGET_S", "200")), time.perf_counter()     # bounded sample: never minutes of CPU
    for it in range(warmup + steps):
        if times and time.perf_counter() - t_start + times[-1] > budget:
            break
        x = torch.randn(bs, 3, 500, 500, generator=g)
        boxes = [random_boxes(rng) for _ in range(bs)]
        t0 = time.perf_counter()
        maps = [otgt.get_heatmaps(b.copy(), t, pad) for b in boxes]
        cm = torch.from_numpy(np.ascontiguousarray(np.stack([c.transpose(2, 0, 1) for c, _, _ in maps]))).float()
        rm = torch.from_numpy(np.ascontiguousarray(np.stack([r.transpose(2, 0, 1) for _, r, _ in maps]))).float()
        t1 = time.perf_counter()
        out = m(x)
        r = ocrit.criterion(out, cm, rm)
        opt.zero_grad()
        out.backward(r["grad"])
        opt.step()
        times.append(time.perf_counter() - t0)
        t_tgt.append(t1 - t0)
    timed = times[warmup:] if len(times) > warmup else times
    steps = len(timed)
    dt = float(np.median(timed))
    # the reference's own target assignment is a four-deep Python loop (dense_overlap.py:30-75): its literal cost for ONE box
    b1 = np.array([[100.0, 120.0, 180.0, 230.0]])
    t0 = time.perf_counter()
    otgt.dense_overlap_loop(-1, -1, 8, 8, 63, 63, t[:, 0], t[:, 1], t[:, 2], t[:, 3], b1[:, 0], b1[:, 1], b1[:, 2], b1[:, 3])
    loop_ms = (time.perf_counter() - t0) * 1e3
    # eval forward of one 500x500 image
    m.eval()
    xe = torch.randn(1, 3, 500, 500, generator=g)
    fw = []
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            m(xe)
            fw.append(time.perf_counter() - t0)
    # eval leg: the 3-scale pyramid of configs[1] on the CPU path, on FRESH weights (the three SGD steps above move the score
    # distribution of the random-weight detector: 45 k survivors and minutes of CPU NMS instead of a few thousand candidates)
    m = o_tame(OracleDetectionModel(num_templates=25), 0).eval()
    img = torch.rand(3, 960, 1280, generator=torch.Generator().manual_seed(0))
    tf = Compose([ToTensor(), Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    et, kept = [], 0
    for it in range(eval_warmup + eval_runs):
        if et and time.perf_counter() - t_start + et[-1] > 1.3 * budget:
            break
        t0 = time.perf_counter()
        d = opyr.get_detections(m, img, t, otgt.RF, tf, prob_thresh=0.6175, nms_thresh=0.3, scales=(-1, 0, 1))
        et.append(time.perf_counter() - t0)
        kept = d.shape[0]
    return {"value": round(bs / dt, 3), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"median of {steps} timed bs={bs} 500x500 training steps after {warmup} warm-up (numpy target assignment "
                      f"{np.median(t_tgt[-steps:]) * 1e3:.0f} ms/step + torch-CPU fp32 fwd/criterion/bwd/SGD); eval leg: median of "
                      f"{max(1, len(et) - eval_warmup)} get_detections runs after {min(eval_warmup, max(0, len(et) - 1))} warm-up on a 1280x960 image, scales (-1,0,1), CPU NMS",
            "ms_per_step": round(dt * 1e3, 1), "targets_ms_per_image": round(float(np.median(t_tgt[-steps:])) * 1e3 / bs, 1),
            "dense_overlap_quad_loop_ms_per_box": round(loop_ms, 1), "eval_forward_500x500_ms": round(float(np.median(fw[1:])) * 1e3, 1),
            "eval_ms_per_image": round(float(np.median(et[eval_warmup:] if len(et) > eval_warmup else et)) * 1e3, 1), "eval_kept": kept,
            "cpu": _cpu_model(), "host_threads": os.cpu_count(), "torch": torch.__version__}


def _profile_file(name):
    """profiles/r06_<name> (this round's pass over the final binary); the round-5 file while that pass has not been taken yet."""
    for rp in ("r06", "r05"):
        p = os.path.join(ROOT, "profiles", f"{rp}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r06_{name}")


PMC_TRAFFIC_FILE = _profile_file("pmc_traffic.json")
PMC_TRAFFIC_FILE_FP32 = _profile_file("pmc_traffic_fp32.json")
PROFILE_STAMP_FILE = _profile_file("stamp.json")


def profile_staleness(identity, traffic_path=None, stamp_path=None):
    """r6: the committed PMC / rocprof files are measurements of ONE binary.  scripts/pmc_traffic.py stamps the traffic file,
    scripts/stamp_profiles.py the kernel-stats summary, with the identity of the library they were taken on (tf_version, digest of its
    sources, sha256 of the .so); a stamp that is missing or differs from the LOADED library's source digest is reported as stale."""
    def stale(path, key=None):
        if not os.path.exists(path):
            return True
        with open(path) as f:
            d = json.load(f)
        lib = d.get(key) if key else d
        return not isinstance(lib, dict) or lib.get("build_id") != identity["build_id"]
    return {"traffic_stale": stale(traffic_path or PMC_TRAFFIC_FILE, "library"), "rocprof_stale": stale(stamp_path or PROFILE_STAMP_FILE),
            "library": {"tf_version": identity["tf_version"], "build_id": identity["build_id"], "so_sha256": identity["so_sha256"][:16]},
            "files": [os.path.relpath(traffic_path or PMC_TRAFFIC_FILE, ROOT), os.path.relpath(ROCPROF_STATS_FILE, ROOT)]}
PMC_PATTERNS = {6: ("conv3x3h_kernel<tf::bf16_t,",), 7: ("conv3x3h_kernel<tf::f16_t,",), 13: ("conv_dma_kernel<tf::bf16_t,",), 12: ("conv_dma_kernel<float,",), 15: ("conv_dma_kernel<tf::f16_t,",),
                14: ("wgrad_dma_kernel",), 16: ("wgrad3x3_kernel",), 17: ("conv_pwx_kernel",), 18: ("wgrad_group_kernel", "wgrad_group_fast_kernel"), 20: ("stem_conv_kernel<tf::bf16_t,",), 21: ("stem_conv_kernel<tf::f16_t,",), 19: ("wgrad3x3_group_kernel",), 23: ("conv_pws_kernel",)}


def pmc_traffic(kind, path=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (scripts/gpu_pmc_bench.sh: FETCH_SIZE x2 + WRITE_SIZE, separate passes; regenerated with the final binary of the round).
    Counters cannot be read from inside the process, so this is the offline measurement, labelled as such.  None when there is no
    pattern for the kind; a RuntimeError when the file exists but holds no kernel of that name any more (a stale file must not
    pass silently: tests/test_host_logic.py::test_bench_reads_committed_pmc_traffic)."""
    path = path or PMC_TRAFFIC_FILE
    pats = PMC_PATTERNS.get(kind)
    if pats is None or not os.path.exists(path):
        return None
    with open(path) as f:
        data = json.load(f)
    rows = [v for k, v in data["kernels"].items() if k.startswith(pats)]
    n = sum(r["launches"] for r in rows)
    if not n:
        raise RuntimeError(f"{path} holds no kernel named {pats}: regenerate it (scripts/gpu_pmc_bench.sh) with the current binary")
    return round(sum(r["hbm_bytes"] * r["launches"] for r in rows) / n)


def pmc_step_traffic(path=None):
    """HBM bytes of ONE whole training step: every kernel of the committed PMC passes (bytes per launch x launches), divided by the steps of
    that run (the fused SGD launches three times per step: the groups with a non-zero learning rate).  None without the file."""
    path = path or PMC_TRAFFIC_FILE
    if not os.path.exists(path):
        return None
    with open(path) as f:
        k = json.load(f)["kernels"]
    sgd = sum(v["launches"] for n, v in k.items() if "sgd_kernel" in n)
    if not sgd:
        return None
    return sum(v["hbm_bytes"] * v["launches"] for v in k.values()) / (sgd / 3.0)


ROCPROF_STATS_FILE = _profile_file("train_bs12_bf16_kernel_stats.csv")


def rocprof_avg_us(kind):
    """Average TRUE kernel duration (microseconds) of the dominant kernel in the committed `rocprofv3 --kernel-trace --stats` summary of
    this same command (`scripts/gpu_job.sh prof train`, regenerated with the final binary of the round): the cross-check of `avg_launch_us`.  The
    HIP-event bracket of bench.py is systematically longer: it spans record -> dispatch -> kernel -> record, i.e. the ~5-6 us
    inter-packet latency of the queue on top of the kernel.  None without a pattern / file; RuntimeError when the file no longer holds
    the kernel (a stale summary must not pass silently)."""
    pats = PMC_PATTERNS.get(kind)
    if pats is None or not os.path.exists(ROCPROF_STATS_FILE):
        return None
    import csv
    calls, total = 0, 0.0
    with open(ROCPROF_STATS_FILE) as f:
        for r in csv.DictReader(f):
            name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            if name.startswith(pats):
                calls += int(r["Calls"]); total += float(r["TotalDurationNs"])
    if not calls:
        raise RuntimeError(f"{ROCPROF_STATS_FILE} holds no kernel named {pats}: regenerate it (scripts/gpu_job.sh prof train) with the current binary")
    return round(total / calls / 1e3, 2)


EPI_NAMES = [(1, "affine"), (2, "res"), (4, "relu"), (8, "stats"), (16, "mask"), (32, "stats2"), (64, "join"), (128, "mask2"), (256, "stats3")]


def write_layer_table(_hip, path, steps, dt):
    """Per layer shape: launches per step, average launch duration (HIP events around every launch), achieved TFLOP/s and GB/s
    (algorithmic), and the two roofline times of that shape: flops / 2.5 PFLOP/s and bytes / 6.3 TB/s (achievable HBM)."""
    rows = (C.c_double * (12 * 256))()
    n = _hip.lib().tf_profile_shapes(rows, 256)
    out = []
    for i in range(n):
        kind, M, N, K, taps, mode, epi, launches, ms, flops, nbytes, xflops = [rows[i * 12 + j] for j in range(12)]
        us = ms * 1e3 / launches
        f1, b1 = flops / launches, nbytes / launches
        t_mfma, t_hbm = f1 / 2.5e15 * 1e6, b1 / 6.3e12 * 1e6
        out.append({"kernel": KIND_NAMES.get(int(kind), str(int(kind))), "op": ["conv", "dgrad", "wgrad"][int(mode)], "M": int(M), "N": int(N), "K": int(K),
                    "taps": int(taps), "epilogue": "+".join(nm for bit, nm in EPI_NAMES if int(epi) & bit) or "-",
                    "launches_per_step": round(launches / steps, 2), "avg_us": round(us, 2), "ms_per_step": round(ms / steps, 3),
                    "tflops": round(f1 / us / 1e6, 1), "executed_tflops": round(xflops / launches / us / 1e6, 1), "gb_s": round(b1 / us / 1e3, 1),
                    "mfma_bound_us": round(t_mfma, 2), "hbm_bound_us": round(t_hbm, 2),
                    "x_over_roofline": round(us / max(t_mfma, t_hbm), 1)})
    out.sort(key=lambda r: -r["ms_per_step"])
    with open(path, "w") as f:
        json.dump({"note": "HIP events around every launch (perturbs the step by ~10 %); roofline = max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s); "
                           "tflops = ALGORITHMIC (2 x MACs of the forward conv the launch belongs to, unpadded channels), executed_tflops = the GEMM the kernel ran "
                           "(a stride-2 data gradient executes its zero-inserted gather at 4x, the stem / heads their padded K / N)",
                   "ms_per_step_with_events": round(dt / steps * 1e3, 3), "shapes": out}, f, indent=1)
    md = os.path.splitext(path)[0] + ".md"
    with open(md, "w") as f:
        f.write("| op | M | N | K | taps | epilogue | launches/step | avg us | ms/step | TFLOP/s (algorithmic) | TFLOP/s (executed) | GB/s | MFMA-bound us | HBM-bound us | x over roofline |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in out:
            f.write("| {op} | {M} | {N} | {K} | {taps} | {epilogue} | {launches_per_step} | {avg_us} | {ms_per_step} | {tflops} | {executed_tflops} | {gb_s} | {mfma_bound_us} | "
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
            outs = model.forward_levels([x for _, x in levels])       # the levels side by side on the model's lanes (evaluation._decode_levels)
            if thr is not None:
                for (s, x), out in zip(levels, outs):
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
        # throughput form = evaluation.get_detections_batch (what evaluate_model.py's image loop amounts to on a list): B images back to
        # back, ONE host synchronisation (the candidate counts), ONE batched NMS over the B segments; rows identical to the per-image loop
        B, tb = 8, []
        dets_b = torch.empty(cap * B, 5, dtype=torch.float64, device=device)
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            count = torch.zeros(1, dtype=torch.int32, device=device)
            marks = torch.zeros(B + 1, dtype=torch.int32, device=device)
            for i in range(B):
                outs = model.forward_levels([x for _, x in levels])
                for (s, x), out in zip(levels, outs):
                    ops.decode_compact(out[0], t_d, masks[s][0], masks[s][1], thr, s, dets_b, count)
                marks[i + 1:i + 2].copy_(count, non_blocking=True)
            offs = marks.tolist()
            cand = dets_b[:offs[-1]]
            keeps = ops.nms_batched(cand[:, :4].contiguous(), cand[:, 4].contiguous(), offs, 0.3)
            res_b = [cand[k].cpu() for k in keeps]
            torch.cuda.synchronize()
            tb.append((time.perf_counter() - t0) / B)
        batched_ok = all(r.shape[0] == n_keep for r in res_b)          # the same image B times: every segment keeps what the single call kept
    ms = float(np.median(times[1:])) * 1e3
    ms_b = float(np.median(tb[1:])) * 1e3
    gflop = 1829.4
    extra = {}
    try:
        extra.update(bench_eval_end_to_end(model, templates, device, runs=max(4, runs // 2)))
    except Exception as e:
        extra["end_to_end_error"] = repr(e)
    try:
        extra.update(bench_eval_fp32(model, levels, t_d, masks, templates, device))
    except Exception as e:
        extra["fp32_error"] = repr(e)
    return {"ms_per_image": round(ms, 3), **extra, "pyramid": "480x640+960x1280+1920x2560", "lanes": max(1, len(model._lanes) + 1), "candidates": n_cand, "kept": n_keep,
            "batched": {"images": B, "ms_per_image": round(ms_b, 3), "same_keeps_as_single": bool(batched_ok), "achieved_tflops": round(gflop / ms_b, 2),
                        "frac_of_bf16_mfma_peak": round(gflop / ms_b / PEAK_TFLOPS["bf16"], 4),
                        "note": "get_detections_batch form: 8 images back to back, one host synchronisation, one batched NMS (ms_per_image above is the "
                                "latency of ONE image with its own synchronisations)"},
            "prob_thresh": round(thr, 5),
            "threshold_note": "calibrated once per run to the 99.5th percentile of the sigmoid scores of the three maps (random weights have no "
                              "WIDER-like sparsity, SURVEY.md 8d); the timed images reuse it",
            "achieved_tflops": round(gflop / ms, 2), "frac_of_bf16_mfma_peak": round(gflop / ms / PEAK_TFLOPS["bf16"], 4)}


def bench_eval_end_to_end(model, templates, device, runs=8):
    """r6: image in -> detections out.  One 1280x960 image on the HOST goes through the product call
    `evaluation.get_detections(..., scales=(-1, 0, 1), pyramid_on_gpu=True)`: the upload of the 3.7 MB image, the three Pillow-exact resizes +
    ToTensor + Normalize on the GPU (tf_image_prepare), three forwards on the lanes, decode, one NMS, the copy of the surviving rows back.
    Inside one constant_weights() session like evaluate_model.py's image loop."""
    from tinyfaces import evaluation, ops, transforms
    model.eval()
    rs = np.random.RandomState(11)
    base = rs.randint(0, 256, (60, 80, 3)).astype(np.uint8)      # blocky random image (pure pixel noise resamples to grey at scale 0.5)
    u8 = np.kron(base, np.ones((16, 16, 1), np.uint8)) ^ rs.randint(0, 32, (960, 1280, 3)).astype(np.uint8)
    img = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)      # what the val loader hands evaluate_model.py: a float CHW tensor in [0, 1] on the host (wider_face.py:224-233)
    tfm = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    with torch.no_grad(), model.constant_weights(reserve=(1, 1920, 2560)):
        levels = evaluation._pyramid_levels(img, (-1, 0, 1), tfm, True, device)
        outs = model.forward_levels([x for _, x in levels])
        allp = torch.cat([torch.sigmoid(o[0, :25]).flatten() for o in outs])
        thr = float(torch.quantile(allp[torch.randperm(allp.numel(), device=device)[:1000000]], 0.995))
        times, kept = [], 0
        for _ in range(runs + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = evaluation.get_detections(model, img, templates, ops.RF, tfm, prob_thresh=thr, nms_thresh=0.3, scales=(-1, 0, 1), device=device, pyramid_on_gpu=True)
            times.append(time.perf_counter() - t0)
            kept = int(d.shape[0])
    return {"end_to_end_ms_per_image": round(float(np.median(times[2:])) * 1e3, 3), "end_to_end_kept": kept,
            "end_to_end_note": "evaluation.get_detections(pyramid_on_gpu=True) on one 1280x960 image tensor held by the host (the val loader's contract): quantise to uint8, upload, GPU resize + normalise of the "
                               "three levels, forwards, decode, NMS, rows back to the host (ms_per_image above starts from resident normalised levels)"}


def bench_eval_fp32(model, levels, t_d, masks, templates, device, runs=5):
    """r6: the same pyramid on the fp32 instantiation -- the path whose NMS-surviving index set is asserted identical
    to the oracle's (tests/test_gpu_fullsize.py); `ms_per_image` is the bf16 path, whose parity bar is the separated-logit fixture of
    tests/test_gpu_fullsize.py::test_get_detections_bf16_fp16_keep_set_on_separated_logits."""
    from tinyfaces import ops
    from tinyfaces.models.model import DetectionModel
    m32 = DetectionModel(num_objects=1, num_templates=25)
    m32.load_state_dict(model.state_dict())
    m32 = m32.set_compute_dtype("fp32").to(device).eval()
    cap = sum((x.shape[2] // 8 + 1) * (x.shape[3] // 8 + 1) for _, x in levels) * 25
    dets = torch.empty(cap, 5, dtype=torch.float64, device=device)
    times, thr, n_keep = [], None, 0
    with torch.no_grad(), m32.constant_weights(reserve=(1, 1920, 2560)):
        for it in range(runs + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            count = torch.zeros(1, dtype=torch.int32, device=device)
            outs = m32.forward_levels([x for _, x in levels])
            if thr is None:
                allp = torch.cat([torch.sigmoid(o[0, :25]).flatten() for o in outs])
                thr = float(torch.quantile(allp[torch.randperm(allp.numel(), device=device)[:1000000]], 0.995))
                continue
            for (s_, x), out in zip(levels, outs):
                ops.decode_compact(out[0], t_d, masks[s_][0], masks[s_][1], thr, s_, dets, count)
            n = int(count.item())
            keep = ops.nms(dets[:n, :4].contiguous(), dets[:n, 4].contiguous(), 0.3)
            res = dets[:n][keep].cpu()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            n_keep = res.shape[0]
    del m32
    torch.cuda.empty_cache()
    ms = float(np.median(times[1:])) * 1e3
    return {"fp32_ms_per_image": round(ms, 3), "fp32_kept": n_keep, "fp32_frac_of_fp32_mfma_peak": round(1829.4 / ms / PEAK_TFLOPS["fp32"], 4),
            "fp32_note": "the same three resident levels on the fp32 parity instantiation (v_mfma_f32_16x16x4_f32): the path with index-exact NMS survivors vs the CPU oracle"}


def bench_eval_forward(model, device, bs=12, runs=10):
    """The forward pass north_star states its MFMA target on, without the training-mode statistics: eval-mode graph (BN folded into
    the conv epilogues: 105 conv launches + stem + heads, nothing else) on a resident bs x 3 x 500 x 500 batch, weights packed once."""
    model.eval()
    x = torch.randn(bs, 3, 500, 500, generator=torch.Generator().manual_seed(3)).to(device)
    with torch.no_grad(), model.constant_weights(reserve=(bs, 500, 500)):
        for _ in range(3):
            model(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(runs):
            model(x)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / runs
    tf_ = FWD_GFLOP_PER_IMG * bs / ms
    return {"ms": round(ms, 3), "img_s": round(bs / ms * 1e3, 1), "achieved_tflops": round(tf_, 2), "frac_of_bf16_mfma_peak": round(tf_ / PEAK_TFLOPS["bf16"], 4),
            "note": f"{runs} eval-mode forwards of a resident {bs}x3x500x500 batch (BN folded, packed weights kept), outside the timed region"}


def bench_fp32_path(device, rank, batch, steps=20, warmup=3):
    """The SAME training step on the fp32 parity path (v_mfma_f32_16x16x4_f32 operands: the instantiation the 1e-3 bar of north_star is
    asserted on at this size, tests/test_gpu_fullsize.py) as a result of its own: >= 20 timed steps, its dominant kernel timed in-run by
    the launches' own time stamps, its fraction of the fp32 MFMA peak (157.3 TFLOP/s) and its committed PMC traffic."""
    from tinyfaces import _hip
    from tinyfaces.datasets.templates import load_templates
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel
    from tinyfaces import ops
    t_d = torch.as_tensor(load_templates(), dtype=torch.float64).to(device)
    torch.manual_seed(0)
    m32 = tame_init_(DetectionModel(num_objects=1, num_templates=25)).set_compute_dtype("fp32")
    eng = TrainEngine(m32, DetectionCriterion(25, seed=rank, lazy_meters=True), lr=1e-4, momentum=0.9, weight_decay=5e-4, device=device)
    b = synthetic_batch(999 + rank, batch, device, t_d)

    def one(i):
        cm, rm = ops.dense_overlap_targets_device(b["boxes"], b["offs"], b["total"], t_d, paste_d=b["paste"], seed=i)
        return eng.step(b["x"], cm, rm)
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    _hip.lib().tf_profile_enable(PROFILE_EVERY)
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _hip.lib().tf_profile_enable(0)
    rows = (C.c_double * (24 * 6))()
    n = _hip.lib().tf_profile_collect(rows, 24)
    prof = [dict(kind=int(rows[i * 6]), launches=int(rows[i * 6 + 1]), ms=rows[i * 6 + 2], flops=rows[i * 6 + 3], bytes=rows[i * 6 + 4]) for i in range(n)]
    del eng, m32
    torch.cuda.empty_cache()
    out = {"img_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "fp32",
           "step_tflops": round(3 * FWD_GFLOP_PER_IMG * batch / (dt * 1e3), 2),
           "frac_of_fp32_mfma_peak": round(3 * FWD_GFLOP_PER_IMG * batch / (dt * 1e3) / PEAK_TFLOPS["fp32"], 4),
           "parity": "per-anchor maps within 1e-3 of the CPU oracle at this size, gradient cosine >= 0.9999 (tests/test_gpu_fullsize.py)",
           "note": f"{steps} steps after {warmup} warm-up of the same workload with fp32 operands (exact fp32 MFMA), outside the bf16 timed region"}
    if prof:
        dom = max(prof, key=lambda r: r["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": KIND_NAMES.get(dom["kind"], str(dom["kind"])), **roof_of(dom["flops"], dom["bytes"], dom["ms"] * 1e-3, PEAK_TFLOPS["fp32"]),
                           "achieved_tflops": round(ach, 2), "traffic": pmc_traffic(dom["kind"], PMC_TRAFFIC_FILE_FP32),
                           "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of `bench.py --dtype fp32` (profiles/r05_pmc_traffic_fp32.json); "
                                           "algorithmic bytes per launch = %d" % round(dom["bytes"] / dom["launches"]),
                           "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2), "launches_sampled": dom["launches"],
                           "launches_per_step": round(dom["launches"] * PROFILE_EVERY / steps, 1),
                           "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
                           "share_of_step": round(dom["ms"] * PROFILE_EVERY / (dt * steps * 1e3), 3)}
    return out


def bench_eval_hard(model, templates, device, runs=5):
    """configs[4], the hard setting: fp16 MFMA convs, a pyramid whose top level has a 5000-px long side (937x1250 + 1875x2500 +
    3750x5000 of a 1875x2500 image), decode, ONE NMS per image; plus the batched multi-scale NMS alone: N = 65 536 boxes of the
    template sizes in one list, and 8 lists of 8 192 in one tf_nms_f64_batched call."""
    from tinyfaces import ops
    prev = model.compute_dtype
    model.set_compute_dtype(torch.float16).eval()
    g = torch.Generator().manual_seed(1)
    levels = [(0.5, torch.randn(1, 3, 937, 1250, generator=g).to(device)), (1, torch.randn(1, 3, 1875, 2500, generator=g).to(device)),
              (2, torch.randn(1, 3, 3750, 5000, generator=g).to(device))]
    t_d = torch.as_tensor(templates, dtype=torch.float64).to(device)
    cap = sum((x.shape[2] // 8 + 1) * (x.shape[3] // 8 + 1) for _, x in levels) * 25
    dets = torch.empty(cap, 5, dtype=torch.float64, device=device)
    masks = {s: [torch.from_numpy(a).to(device) for a in ops.template_masks(templates, s, (x.shape[3] + 7) // 8, "w")] for s, x in levels}
    thr, times, n_cand, n_keep = None, [], 0, 0
    try:
        with torch.no_grad(), model.constant_weights(reserve=(1, 3750, 5000)):
            for it in range(runs + 2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                count = torch.zeros(1, dtype=torch.int32, device=device)
                outs = model.forward_levels([x for _, x in levels])
                if thr is not None:
                    for (s, x), out in zip(levels, outs):
                        ops.decode_compact(out[0], t_d, masks[s][0], masks[s][1], thr, s, dets, count)
                if thr is None:
                    allp = torch.cat([torch.sigmoid(o[0, :25]).flatten() for o in outs])
                    thr = float(torch.quantile(allp[torch.randperm(allp.numel(), device=device)[:2000000]], 0.999))
                    del outs, allp
                    continue
                n = int(count.item())
                keep = ops.nms(dets[:n, :4].contiguous(), dets[:n, 4].contiguous(), 0.3)
                res = dets[:n][keep].cpu()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                n_cand, n_keep = n, res.shape[0]
            # throughput form (get_detections_batch): 4 images back to back, one synchronisation, one batched NMS
            B, tb = 4, []
            dets_b = torch.empty(cap * B, 5, dtype=torch.float64, device=device)
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                count = torch.zeros(1, dtype=torch.int32, device=device)
                marks = torch.zeros(B + 1, dtype=torch.int32, device=device)
                for i in range(B):
                    outs = model.forward_levels([x for _, x in levels])
                    for (s, x), out in zip(levels, outs):
                        ops.decode_compact(out[0], t_d, masks[s][0], masks[s][1], thr, s, dets_b, count)
                    marks[i + 1:i + 2].copy_(count, non_blocking=True)
                offs_b = marks.tolist()
                cand = dets_b[:offs_b[-1]]
                keeps = ops.nms_batched(cand[:, :4].contiguous(), cand[:, 4].contiguous(), offs_b, 0.3)
                res_b = [cand[k].cpu() for k in keeps]
                torch.cuda.synchronize()
                tb.append((time.perf_counter() - t0) / B)
            batched_ok = all(r.shape[0] == n_keep for r in res_b)
            del outs, dets_b, cand
    finally:
        model.set_compute_dtype(prev)
    ms = float(np.median(times[1:])) * 1e3
    ms_b = float(np.median(tb[1:])) * 1e3
    gflop = FWD_GFLOP_PER_IMG * (937 * 1250 + 1875 * 2500 + 3750 * 5000) / 250000.0      # conv FLOPs scale with the pixel count
    # the NMS alone at the cfg5 sizes
    rng = np.random.RandomState(3)

    def boxes(n, w, h):
        t = templates[rng.randint(0, templates.shape[0], n)]
        bw, bh = t[:, 2] - t[:, 0] + 1, t[:, 3] - t[:, 1] + 1
        cx, cy = rng.uniform(0, w, n), rng.uniform(0, h, n)
        return np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1), rng.randn(n)
    b1, s1 = boxes(65536, 5000, 3750)
    B1, S1 = torch.from_numpy(b1).to(device), torch.from_numpy(s1).to(device)
    b8, s8 = boxes(65536, 2500, 1875)
    B8, S8 = torch.from_numpy(b8).to(device), torch.from_numpy(s8).to(device)
    offs = [8192 * i for i in range(9)]

    def t_of(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3, r
    ms1, k1 = t_of(lambda: ops.nms(B1, S1, 0.3))
    ms8, k8 = t_of(lambda: ops.nms_batched(B8, S8, offs, 0.3))
    ms8_loop, _ = t_of(lambda: [ops.nms(B8[a:b].contiguous(), S8[a:b].contiguous(), 0.3) for a, b in zip(offs, offs[1:])])
    return {"dtype": "f16", "pyramid": "937x1250+1875x2500+3750x5000", "ms_per_image": round(ms, 3), "candidates": n_cand, "kept": n_keep,
            "achieved_tflops": round(gflop / ms, 2), "frac_of_f16_mfma_peak": round(gflop / ms / PEAK_TFLOPS["bf16"], 4),
            "arena_gb": round(_arena_gb(model, 3750, 5000), 2),
            "batched": {"images": B, "ms_per_image": round(ms_b, 3), "same_keeps_as_single": bool(batched_ok), "achieved_tflops": round(gflop / ms_b, 2),
                        "frac_of_f16_mfma_peak": round(gflop / ms_b / PEAK_TFLOPS["bf16"], 4)},
            "nms_65536": {"ms": round(ms1, 3), "kept": int(k1.numel()), "iou_evals_per_s": round(65536 * 65535 / 2 / (ms1 * 1e-3), 0)},
            "nms_batched_8x8192": {"ms": round(ms8, 3), "ms_as_8_calls": round(ms8_loop, 3), "kept": int(sum(k.numel() for k in k8))}}


def _arena_gb(model, H, W):
    from tinyfaces import _hip
    return _hip.lib().tf_detnet_workspace_bytes(_hip.TF_F16, 1, H, W, model.num_out, 0) / 2**30


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE / RANK in the environment): start the N ranks ourselves --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same
    arguments>`, one process per GPU over RCCL; rank 0's JSON line passes through on stdout, the exit code is the job's."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    share = os.environ.get("TINYFACES_BENCH_SHARE_GPU") == "1"
    if have == 0:
        sys.stderr.write(f"bench.py --gpus {n}: no GPU is visible to this process (the HIP path has no CPU fallback)\n")
        return 3
    if have < n and not share:
        sys.stderr.write(f"bench.py --gpus {n}: only {have} GPU(s) visible; one process per GPU needs {n} "
                         "(TINYFACES_BENCH_SHARE_GPU=1 runs the ranks over gloo on shared devices: functional test only)\n")
        return 3
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def roof_of(flops, nbytes, seconds, peak_tflops):
    """Which roof binds a launch: the one whose minimum time is larger -- t_mfma = flops / peak, t_hbm = bytes / 8 TB/s,
    both from the ALGORITHMIC work of the launch.  Returns the roofline fields expressed against THAT roof plus both fractions."""
    t_m, t_h = flops / (peak_tflops * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
    frac_m, frac_h = t_m / seconds, t_h / seconds
    if t_h >= t_m:
        return {"bound": "hbm", "achieved": round(nbytes / seconds / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(frac_h, 4),
                "frac_mfma": round(frac_m, 4), "frac_hbm": round(frac_h, 4)}
    return {"bound": "mfma", "achieved": round(flops / seconds / 1e12, 2), "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(frac_m, 4),
            "frac_mfma": round(frac_m, 4), "frac_hbm": round(frac_h, 4)}


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
    ap.add_argument("--no-fp32-path", action="store_true", help="skip the fp32 parity-path throughput leg (a second model + arena, ~1 s of GPU time)")
    ap.add_argument("--no-eval-hard", action="store_true", help="skip the configs[4] leg (5000-px fp16 pyramid + batched NMS)")
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline with the full BASELINE.md protocol (3 + 5 training steps, 3 + 20 images) instead of the bounded sample")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))      # `python bench.py --gpus N`: this process becomes the launcher of N ranks

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
        if args.layer_table:              # per-shape table of the pyramid's conv launches (HIP events around every launch)
            _hip.lib().tf_profile_enable(1)
        t0 = time.perf_counter()
        ev = bench_eval(model.to(device), templates, device, runs=10)
        if args.layer_table:
            _hip.lib().tf_profile_enable(0)
            _hip.lib().tf_profile_collect((C.c_double * (6 * 24))(), 24)
            write_layer_table(_hip, args.layer_table, 12, time.perf_counter() - t0)      # 12 images: 10 timed + calibration + warm-up
        os.write(json_fd, (json.dumps({"eval": ev}) + "\n").encode())
        return
    crit = DetectionCriterion(25, seed=rank, lazy_meters=True)
    eng = TrainEngine(model, crit, lr=1e-4, momentum=0.9, weight_decay=5e-4, device=device)

    if os.environ.get("TINYFACES_BENCH_SKIP_ALLREDUCE"):      # diagnostic: the process group exists, the steps never use it
        eng.skip_allreduce = True
    pool = [synthetic_batch(1000 * s + rank, args.batch, device, t_d) for s in range(4)]

    raw = None
    if args.with_augment:
        from tinyfaces.datasets import augment as aug
        from tinyfaces.datasets.synthetic import random_boxes
        gi = torch.Generator().manual_seed(7 + rank)
        raw = [(torch.randint(0, 256, (768, 1024, 3), dtype=torch.uint8, generator=gi).to(device),
                random_boxes(np.random.RandomState(50 + k + 100 * rank)) * np.array([1024 / 500, 768 / 500] * 2)) for k in range(args.batch)]
        arng = np.random.RandomState(rank)

    # --with-augment builds the batch (pixels + targets) on an input stream of its own, like the product loader does
    # (tinyfaces/datasets/wider_face.py: the collate runs on `_in_stream`, the training stream waits for it).  The resident-input bench
    # line assigns its targets on the training stream (rounds 1-2); TINYFACES_BENCH_TARGETS_STREAM=1 moves them to an input stream
    # too (with the host a step ahead they then run beside the previous backward pass: within noise, A/B r3).
    in_stream = torch.cuda.Stream(device=device) if (raw is not None or os.environ.get("TINYFACES_BENCH_TARGETS_STREAM")) else None

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

    # diagnostic: N more streams with a token amount of work per step, forked from / joined to the training stream --
    # how many HIP streams one process can keep busy before they start sharing hardware queues
    extra_streams = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get("TINYFACES_BENCH_EXTRA_STREAMS", "0")))]
    token = torch.zeros(64, device=device)

    def step(i):
        if extra_streams:
            cur = torch.cuda.current_stream()
            for s in extra_streams:
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    token.add_(1.0)
            out = step_(i)
            for s in extra_streams:
                cur.wait_stream(s)
            return out
        return step_(i)

    def step_(i):
        if raw is not None:               # the real input pipeline of a training step, minus the JPEG decode
            x, cm, rm = build_batch(i)
            cur = torch.cuda.current_stream()
            cur.wait_stream(in_stream)
            for t in (x, cm, rm):
                t.record_stream(cur)
            return eng.step(x, cm, rm)
        b = pool[i % len(pool)]
        if in_stream is None:
            cm, rm = ops.dense_overlap_targets_device(b["boxes"], b["offs"], b["total"], t_d, paste_d=b["paste"], seed=i * world + rank)
            return eng.step(b["x"], cm, rm)
        with torch.cuda.stream(in_stream):
            cm, rm = ops.dense_overlap_targets_device(b["boxes"], b["offs"], b["total"], t_d, paste_d=b["paste"], seed=i * world + rank)
        cur = torch.cuda.current_stream()
        cur.wait_stream(in_stream)
        cm.record_stream(cur); rm.record_stream(cur)
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
    host_n, host_dt = min(4, args.steps), 0.0
    for i in range(args.steps):
        loss2 = step(args.warmup + i)
        if i + 1 == host_n:
            host_dt = time.perf_counter() - t0          # the host has ENQUEUED the first steps here (no synchronisation inside a step; later
                                                        # steps block on the full hardware queue, so only the first few show the host's own cost)
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
            ones = torch.ones(1, device=device)
            dist.all_reduce(ones)                     # what the collective library itself saw: the sum of one 1 per participating rank
            comm = {"ranks": int(ones.item()), "backend": dist.get_backend(), "gradient_bytes": nbytes, "allreduce_ms_standalone": round(t_ar * 1e3, 3),
                    "bus_gb_s": round(2 * (world - 1) / world * nbytes / t_ar / 1e9, 1),
                    "ms_per_step_without_exchange": round(t_nc * 1e3, 3), "exposed_ms_per_step": round(exposed * 1e3, 3),
                    "hidden_fraction": round(max(0.0, 1.0 - exposed / t_ar), 3) if t_ar > 0 else None,
                    "buckets": len(eng._overlap["ranges"]) if eng._overlap else None}
        except Exception as e:      # the measurement is a bonus: report the failure, keep the headline
            comm = {"error": repr(e)}

    def collect():
        rows = (C.c_double * (24 * 6))()
        n = _hip.lib().tf_profile_collect(rows, 24)
        prof = [dict(kind=int(rows[i * 6]), launches=int(rows[i * 6 + 1]), ms=rows[i * 6 + 2], flops=rows[i * 6 + 3], bytes=rows[i * 6 + 4], xflops=rows[i * 6 + 5])
                for i in range(n)]
        srows = (C.c_double * (12 * 256))()
        ns = _hip.lib().tf_profile_shapes(srows, 256)
        shapes = [dict(kind=int(srows[i * 12]), mode=int(srows[i * 12 + 5]), launches=srows[i * 12 + 7], ms=srows[i * 12 + 8], flops=srows[i * 12 + 9],
                       bytes=srows[i * 12 + 10], xflops=srows[i * 12 + 11]) for i in range(ns)]
        return prof, shapes

    prof, shapes = collect()
    if rank != 0:
        return
    # ---- outside the timed region (N = 1): (a) the forward pass alone (north_star states its MFMA target on it), (b) three steps with
    #      EVERY MFMA launch bracketed and the weight gradients on the caller's stream, so that per-kernel times are free of
    #      two-stream contention and add up to less than the step they were taken from
    extra = {}
    if world == 1 and not args.no_profile and not args.layer_table:       # (--layer-table keeps the shapes of the TIMED region)
        try:
            b = pool[0]
            m = eng.model
            for _ in range(3):
                m._run_forward(b["x"], training=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m._run_forward(b["x"], training=True)
            e1.record(); torch.cuda.synchronize()
            extra["fwd_ms"] = e0.elapsed_time(e1) / 10
            m.single_stream = True
            step(0); torch.cuda.synchronize()
            _hip.lib().tf_profile_enable(1)
            t1 = time.perf_counter()
            for i in range(3):
                step(1 + i)
            torch.cuda.synchronize()
            extra["ss_ms_per_step"] = (time.perf_counter() - t1) / 3 * 1e3
            _hip.lib().tf_profile_enable(0)
            extra["ss_prof"], extra["ss_shapes"] = collect()
        except Exception as e:
            extra["error"] = repr(e)
        finally:
            _hip.lib().tf_profile_enable(0)
            eng.model.single_stream = False
    if args.layer_table:
        write_layer_table(_hip, args.layer_table, args.steps, dt)
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    out = {"metric": "train img/s @ 500x500 bs=12/GPU", "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "host_enqueue_ms_per_step": round(host_dt / host_n * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "configs[2]: bs=12/GPU synthetic 500x500 crops + random boxes; dense_overlap targets on GPU, "
                                  "ResNet-101 hybrid-head fwd, criterion, bwd, fused SGD" + (", RCCL grad all-reduce" if world > 1 else "") +
                                  (" + GPU augmentation of 768x1024 uint8 images (8f.1) every step" if args.with_augment else ""),
                      "global_batch": args.batch * world, "image": "500x500", "templates": 25, "parallelism": f"dp{world}",
                      "weights": "random init (tamed kaiming), fp32 master + " + args.dtype + " MFMA operands"},
           "precision_note": "bf16 MFMA operands, fp32 accumulation and master weights (bars at this size: forward maps <= 1.8e-2 of a 1.19 range, gradient "
                             "cosine >= 0.90 except <= 2 layer-1 BN sums >= 0.85: tests/test_gpu_fullsize.py); the equal-precision number of the reference's "
                             "fp32 arithmetic is `fp32_path` in this line (1e-3 bar)",
           "loss": {"cls": round(loss_v[0], 3), "reg": round(loss_v[1], 3)},
           "step_tflops": round(3 * FWD_GFLOP_PER_IMG * args.batch / ms_per_step, 2)}
    if prof:
        dom = max(prof, key=lambda r: r["ms"])
        peak = PEAK_TFLOPS["bf16" if dom["kind"] in BF16_KINDS else "fp32"]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = roof_of(dom["flops"], dom["bytes"], dom["ms"] * 1e-3, peak)
        out["roofline"] = {"kernel": KIND_NAMES.get(dom["kind"], str(dom["kind"])), **roof,
                           "bound_note": "bound = argmax(algorithmic flops / MFMA peak, algorithmic bytes / 8 TB/s) of the kernel's launches in the timed region; "
                                         "achieved / peak / frac are stated against that roof, frac_mfma and frac_hbm against both",
                           "achieved_tflops": round(ach, 2), "peak_tflops": peak,
                           "arithmetic_intensity_flop_per_byte": round(dom["flops"] / max(dom["bytes"], 1.0), 1), "machine_balance_flop_per_byte": round(peak * 1e12 / (PEAK_HBM_GBS * 1e9), 1),
                           "traffic": pmc_traffic(dom["kind"]),
                           **{k: v for k, v in profile_staleness(_hip.identity()).items() if k in ("traffic_stale", "rocprof_stale")},
                           "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this command (%s); traffic_stale / rocprof_stale: "
                                           "the file's library stamp differs from the loaded library (profiles_identity); "
                                           "algorithmic bytes per launch = %d" % (os.path.relpath(PMC_TRAFFIC_FILE, ROOT), round(dom["bytes"] / dom["launches"])),
                           "launches_sampled": dom["launches"], "launches_per_step": round(dom["launches"] * PROFILE_EVERY / args.steps, 1),
                           "sampling": f"HIP events around 1 launch in {PROFILE_EVERY} over the timed region (weight gradients run concurrently on a second stream)",
                           "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                           "rocprof_avg_launch_us": rocprof_avg_us(dom["kind"]),
                           "frac_rocprof_clock": (lambda us: roof_of(dom["flops"] / dom["launches"], dom["bytes"] / dom["launches"], us * 1e-6, peak)["frac"] if us else None)(rocprof_avg_us(dom["kind"])),
                           "rocprof_note": "true kernel duration in the committed rocprofv3 --kernel-trace --stats summary of this command "
                                           "(%s); the HIP-event bracket adds the queue's inter-packet latency (~5-6 us)" % os.path.relpath(ROCPROF_STATS_FILE, ROOT),
                           "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
                           "executed_gflop_per_launch": round(dom["xflops"] / dom["launches"] / 1e9, 3),
                           "flops_note": "achieved / frac count ALGORITHMIC flops: 2 x the MACs of the forward convolution a launch belongs to on unpadded channels "
                                         "(SURVEY.md 8d); executed = the 2*M*N*K of the GEMM the kernel ran (stride-2 data gradients at 4x, padded stem / head channels)",
                           "algorithmic_gb_s": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                           "share_of_timed_region": round(dom["ms"] * PROFILE_EVERY / (dt * 1e3), 3)}
        fam = [r for r in prof if r["kind"] in (13, 23)]
        if dom["kind"] in (13, 23) and len(fam) == 2:
            # r5: the short-K / large-M pointwise launches (the launches of this class nearest to the HBM roof) left conv_dma for conv_pws; the two
            # together are the launches `roofline` of rounds 1-4 was quoted on
            out["roofline"]["with_conv_pws"] = {**{k: v for k, v in roof_of(sum(r["flops"] for r in fam), sum(r["bytes"] for r in fam), sum(r["ms"] for r in fam) * 1e-3, peak).items()
                                                   if k in ("bound", "achieved", "unit", "frac", "frac_mfma", "frac_hbm")},
                                                "note": "conv_dma + conv_pws launches of the timed region together: the kernel class `roofline` was quoted on through round 4 "
                                                        "(conv_pws took its 14 launches per step with the most bytes per launch)"}
        stb = pmc_step_traffic() if args.dtype == "bf16" and world == 1 else None
        if stb:       # the step as a whole against the OTHER roof: a training-mode-BN step at bs = 12 moves every activation several times
            out["roofline_step"] = {"bound": "hbm", "traffic_gb_per_step": round(stb / 1e9, 2), "achieved": round(stb / (ms_per_step * 1e-3) / 1e9, 1),
                                    "peak": 8000.0, "unit": "GB/s", "frac": round(stb / (ms_per_step * 1e-3) / 8e12, 4),
                                    "ms_per_step_at_6300_gb_s": round(stb / 6.3e12 * 1e3, 2),
                                    "mfma_frac_of_step": round(3 * FWD_GFLOP_PER_IMG * args.batch / ms_per_step / PEAK_TFLOPS["bf16"], 4),
                                    "traffic_stale": profile_staleness(_hip.identity())["traffic_stale"],
                                    "note": "all kernels of a step, HBM bytes from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, "
                                            "%s) over this run's step time; the step is nearer to the HBM roof than to the MFMA roof" % os.path.relpath(PMC_TRAFFIC_FILE, ROOT)}
        out["kernels"] = [{"kernel": KIND_NAMES.get(r["kind"], str(r["kind"])), "launches_per_step": round(r["launches"] * PROFILE_EVERY / args.steps, 1),
                           "ms_per_step": round(r["ms"] * PROFILE_EVERY / args.steps, 3),
                           "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1), "gb_s": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1),
                           **{k: v for k, v in roof_of(r["flops"], r["bytes"], r["ms"] * 1e-3, PEAK_TFLOPS["bf16" if r["kind"] in BF16_KINDS else "fp32"]).items()
                              if k in ("bound", "frac_mfma", "frac_hbm")}} for r in prof]
        fw = [r for r in shapes if r["mode"] == 0 and r["kind"] == dom["kind"]]
        if fw:          # the forward convs of the same kernel (no second stream is active during the forward pass)
            f_ms, f_fl = sum(r["ms"] for r in fw), sum(r["flops"] for r in fw)
            out["roofline"]["forward_convs"] = {"achieved": round(f_fl / (f_ms * 1e-3) / 1e12, 2), "frac": round(f_fl / (f_ms * 1e-3) / 1e12 / peak, 4),
                                                "ms_per_step": round(f_ms * PROFILE_EVERY / args.steps, 3)}
        if "fwd_ms" in extra:       # north_star: ">= 40 % MFMA peak on ResNet-101 forward at 500x500 bs=12" -- the whole training-mode forward pass
            tf_ = FWD_GFLOP_PER_IMG * args.batch / extra["fwd_ms"]
            out["roofline"]["forward_pass"] = {"ms": round(extra["fwd_ms"], 3), "achieved": round(tf_, 2), "frac": round(tf_ / peak, 4),
                                               "note": "10 training-mode forwards (batch-stat BN) of the timed batch, outside the timed region"}
        if "ss_prof" in extra:
            out["kernels_single_stream"] = {
                "ms_per_step": round(extra["ss_ms_per_step"], 3),
                "note": "3 extra steps outside the timed region, weight gradients on the caller's stream, HIP events around EVERY MFMA launch: no two-stream contention, the rows add up to less than ms_per_step",
                "kernels": [{"kernel": KIND_NAMES.get(r["kind"], str(r["kind"])), "launches_per_step": round(r["launches"] / 3, 1), "ms_per_step": round(r["ms"] / 3, 3),
                             "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1), "frac": round(r["flops"] / (r["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS["bf16" if r["kind"] in BF16_KINDS else "fp32"], 4)}
                            for r in extra["ss_prof"]]}
        if "error" in extra:
            out["kernels_single_stream"] = {"error": extra["error"]}
    out["profiles_identity"] = profile_staleness(_hip.identity())
    if comm is not None:
        out["allreduce"] = comm
    if world == 1 and args.dtype == "bf16" and not args.no_fp32_path:
        try:
            out["fp32_path"] = bench_fp32_path(device, rank, args.batch)
        except Exception as e:
            out["fp32_path"] = {"error": repr(e)}
    if world == 1 and not args.no_eval:
        try:
            out["eval_forward_bs12_500"] = bench_eval_forward(model, device, bs=args.batch)
        except Exception as e:
            out["eval_forward_bs12_500"] = {"error": repr(e)}
        try:
            out["eval"] = bench_eval(model, templates, device)
        except Exception as e:   # the headline number must still be printed
            out["eval"] = {"error": repr(e)}
    if world == 1 and not args.no_eval and not args.no_eval_hard:
        try:
            out["eval_hard"] = bench_eval_hard(model, templates, device)
        except Exception as e:
            out["eval_hard"] = {"error": repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(warmup=3, steps=5, eval_warmup=3, eval_runs=20) if args.cpu_full else cpu_baseline()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
