"""-m gpu: the data-parallel path on a 1-GPU box: two gloo ranks sharing cuda:0 run the real
TrainEngine (bucketed gradient exchange driven by the executor's gradient-ready events) and must take the SAME steps as a
single process that averages the gradients of the same two micro-batches; plus the property the overlapped exchange rests on:
a bucket's slice of the flat gradient is FINAL when its event fires."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from gpu_util import report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 3


@pytest.mark.parametrize("ranks", [2, 4])
def test_two_rank_engine_equals_single_process_gradient_average(tmp_path, ranks):
    """`ranks` gloo ranks sharing cuda:0 (rank r trains on micro-batch r % 2) against ONE process that sums the gradients of the same micro-batches.
    Four ranks put more than two contributions into every bucket, including the buckets whose boundaries fall inside a
    weight-gradient group of layer 3."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from tinyfaces import _hip, ops
    golden = os.path.join(ROOT, "tests", "golden", "trainer.npz")
    out = str(tmp_path / "rank0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29539 + ranks), os.path.join(ROOT, "tests", "dist_worker.py"), golden, out, str(STEPS)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    dist = np.load(out)
    # r6: with more than one rank the engine's DEFAULT is the native RCCL hook; two ranks sharing one device are refused by RCCL ("Duplicate GPU
    # detected"), every rank sees the refusal (MIN over the gloo group) and the engine falls back to the torch.distributed callback -- once
    native_now, fallbacks = [int(v) for v in dist["native"]]
    assert native_now == 0 and fallbacks == 1, (native_now, fallbacks)
    dist = [dist[f"arr_{i}"] for i in range(STEPS)]

    # the same two micro-batches in ONE process: two replicas, gradients summed, the fused SGD folds the 1/2 in (engine.py)
    prev = _hip.lib().tf_get_stat_rows()
    try:
        _hip.lib().tf_set_stat_rows(0)
        reps = []
        for r_ in range(2):
            m, c, batches = dist_worker.build(golden)
            m = m.cuda().train()
            flat = m.flatten_parameters()
            reps.append(dict(m=m, c=c, flat=flat, mom=torch.zeros_like(flat), batch=[t.cuda() for t in batches[r_]]))
        groups = reps[0]["m"].group_ranges()
        worst = []
        for s in range(STEPS):
            grads = []
            for rp in reps:
                m, c = rp["m"], rp["c"]
                img, cm, rm = rp["batch"]
                m._sync_tables(img.device)
                o = m._run_forward(img, training=True)
                _, g, _ = ops.criterion_fwd_bwd(o, cm.clone(), rm, c.n_templates, c.reg_weight, c.ohem_thresh, c.max_pos, c.max_neg,
                                                c._pos_keep, c._neg_keep, c._next_seed())
                grads.append(m._run_backward(img, g, persistent=True).clone())
            gsum = (grads[0] + grads[1]) * (ranks // 2)               # every micro-batch is trained by ranks / 2 ranks
            for rp in reps:
                for a, b, mult in groups:
                    if mult != 0.0:
                        ops.sgd_step(rp["flat"][a:b], gsum[a:b], rp["mom"][a:b], 1e-4 * mult, 0.9, 5e-4, 1.0 / ranks)
            torch.cuda.synchronize()
            ref = reps[0]["flat"].cpu().numpy()
            d = np.abs(dist[s] - ref)
            worst.append(float(d.max() / (np.abs(ref).max() + 1e-30)))
    finally:
        _hip.lib().tf_set_stat_rows(prev if prev <= 16 else 0)
    report(f"dist_{ranks}_ranks_vs_single", worst_rel=str([f"{w:.2e}" for w in worst]))
    # step 1: only the fp32-atomic summation order of the weight gradients differs between two runs (1e-7); later steps amplify it
    # through batch-statistics BN on these 2-image batches exactly as between two single-process runs (test_gpu_model.py)
    assert all(np.isfinite(d).all() for d in dist) and all(np.isfinite(w) for w in worst), worst
    assert worst[0] < 1e-5, worst
    assert worst[-1] < 1e-2, worst


def test_gradient_bucket_slices_are_final_when_their_event_fires():
    """The overlapped all-reduce (engine._allreduce) reads bucket k's slice of the flat gradient on a communication stream that
    waits for event k only.  Snapshot every slice on such a stream while the backward pass is still running (a long sleep kernel
    in front keeps the GPU behind the host, so every wait is enqueued before its event can fire) and compare with the final
    gradient: bit-equal, for the ~10 MB buckets the engine uses."""
    import ctypes as C
    from oracle.model import OracleDetectionModel, tame_init_
    from tinyfaces import _hip
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
    m = m.cuda().set_compute_dtype(torch.bfloat16).train()
    flat = m.flatten_parameters()
    firsts = TrainEngine.auto_first_blocks(m._segments, flat.numel(), 10)
    ranges = [r for r in TrainEngine.bucket_ranges(m._segments, flat.numel(), firsts) if r[2] > r[1]]
    assert len(ranges) >= 8 and ranges[-1][0] == -1
    x = torch.randn(4, 3, 224, 256, generator=torch.Generator().manual_seed(3)).cuda()
    evs = [torch.cuda.Event() for _ in ranges]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    blocks = (C.c_int * len(ranges))(*[r[0] for r in ranges])
    handles = (C.c_void_p * len(ranges))(*[int(e.cuda_event) for e in evs])
    snaps = []
    try:
        m._grad_events = (blocks, handles, len(ranges))
        for rep in range(3):
            m._sync_tables(x.device)
            out = m._run_forward(x, training=True)
            gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(rep)).cuda()
            torch.cuda._sleep(int(3e8))                                  # ~0.15 s: the whole backward pass queues up behind it
            gflat = m._run_backward(x, gout, persistent=True)
            streams = [torch.cuda.Stream() for _ in ranges]
            cur = []
            for (blk, s, e), ev, st in zip(ranges, evs, streams):
                with torch.cuda.stream(st):
                    st.wait_event(ev)
                    cur.append(gflat[s:e].clone())
            torch.cuda.synchronize()
            snaps.append([bool(torch.equal(c_, gflat[s:e])) and bool(c_.abs().sum() > 0) for c_, (blk, s, e) in zip(cur, ranges)])
    finally:
        m._grad_events = None
    report("grad_slice_finality", buckets=len(ranges), mb=str([round((e - s) * 4 / 2**20, 1) for _, s, e in ranges]), final=str(snaps))
    assert all(all(s) for s in snaps), snaps


def test_native_exchange_one_rank_communicator_and_hook():
    """include/tinyfaces_hip.h tf_comm_* / tf_allreduce_bucket / tf_comm_allreduce_hook (csrc/comm.hip) on the one GPU of the box: a 1-rank RCCL
    communicator (RCCL refuses two ranks on one device, profiles/r03_dist_smoke.txt) -- the sum over one rank is the identity, so the test pins
    the plumbing: librccl resolved at run time, the collective ordered behind the carrying stream, the hook called once per registered
    bucket from inside the backward enqueue with NO Python in between, tf_comm_join, and gradients equal to a run without hooks."""
    import ctypes as C
    from oracle.model import OracleDetectionModel, tame_init_
    from tinyfaces import _hip
    from tinyfaces._hip import lib
    from tinyfaces.engine import TrainEngine
    from tinyfaces.models.model import DetectionModel
    assert lib().tf_comm_available() == 1
    ident = C.create_string_buffer(_hip.TF_COMM_ID_BYTES)
    assert lib().tf_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    assert lib().tf_comm_init(ident, 0, 1, C.byref(comm)) == 0
    try:
        assert lib().tf_comm_world(comm) == 1 and lib().tf_comm_rank(comm) == 0
        # a bucket behind a busy stream
        side = torch.cuda.Stream()
        buf = torch.zeros(1 << 20, device="cuda")
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(5e7))
            buf.fill_(3.0)                                                   # the "gradient kernel" the collective must wait for
        assert lib().tf_allreduce_bucket(comm, buf.data_ptr(), buf.numel(), side.cuda_stream) == 0
        assert lib().tf_comm_join(comm, torch.cuda.current_stream().cuda_stream) == 0
        total = float(buf.sum())                                             # (on the current stream: behind the join)
        assert total == 3.0 * buf.numel()
        # the hook inside the executor
        m = DetectionModel(num_templates=25)
        m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
        m = m.cuda().set_compute_dtype(torch.bfloat16).train()
        flat = m.flatten_parameters()
        firsts = TrainEngine.auto_first_blocks(m._segments, flat.numel(), 10)
        ranges = [r for r in TrainEngine.bucket_ranges(m._segments, flat.numel(), firsts) if r[2] > r[1]]
        n = len(ranges)
        x = torch.randn(2, 3, 160, 192, generator=torch.Generator().manual_seed(4)).cuda()
        m._sync_tables(x.device)
        lib().tf_set_stat_rows(0)
        out = m._run_forward(x, training=True)
        ref = m._run_backward(x, torch.ones_like(out), persistent=True).clone()
        blocks = (C.c_int * n)(*[r[0] for r in ranges])
        start, end = (C.c_int64 * n)(*[r[1] for r in ranges]), (C.c_int64 * n)(*[r[2] for r in ranges])
        plan = _hip.CommPlan()
        plan.comm, plan.grad_flat, plan.n = comm, m._grad_flat_persistent.data_ptr(), n
        plan.blocks, plan.start, plan.end = C.cast(blocks, C.POINTER(C.c_int)), C.cast(start, C.POINTER(C.c_int64)), C.cast(end, C.POINTER(C.c_int64))
        m._grad_events = (blocks, (C.c_void_p * n)(*([None] * n)), n)
        m._grad_callback = C.cast(lib().tf_comm_allreduce_hook, C.c_void_p)
        m._grad_callback_user = C.cast(C.pointer(plan), C.c_void_p)
        try:
            m._run_forward(x, training=True)
            g = m._run_backward(x, torch.ones_like(out), persistent=True)
            assert plan.rc == 0 and plan.issued == n, (plan.rc, plan.issued, n)
            assert lib().tf_comm_join(comm, torch.cuda.current_stream().cuda_stream) == 0
            torch.cuda.synchronize()
        finally:
            m._grad_events = None; m._grad_callback = None; m._grad_callback_user = None
            lib().tf_set_stat_rows(8)
        rel = float((g - ref).abs().max() / ref.abs().max())
        report("native_exchange_1rank", buckets=n, rel=rel)
        assert rel < 1e-3
    finally:
        lib().tf_comm_destroy(comm)


def test_native_exchange_falls_back_to_torch_distributed_on_a_failed_bucket(tmp_path):
    """r6: the C hook is the default exchange for world > 1 and any non-zero rc must end in the torch.distributed exchange, not
    in a lost bucket.  A 1-rank RCCL group; TINYFACES_COMM_FAIL_BUCKET=1 makes tf_comm_allreduce_hook refuse bucket 1: the engine reduces it
    itself behind the join, leaves the native exchange (native_fallbacks == 1, the ctypes callback from step 2 on) and ends where an engine that
    used torch.distributed all along ends."""
    out = str(tmp_path / "nf.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", TINYFACES_COMM_FAIL_BUCKET="1", TINYFACES_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "native-fallback", out], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = np.load(out)
    assert int(d["a_native_at_start"]) == 1 and int(d["a_native_after_step1"]) == 0 and int(d["a_fallbacks"]) == 1
    assert int(d["b_native_at_start"]) == 0 and int(d["b_fallbacks"]) == 0
    assert "not issued" in r.stdout                                         # the engine says what happened, once
    rel = float(np.abs(d["a_flat"] - d["b_flat"]).max() / np.abs(d["b_flat"]).max())
    report("native_exchange_fallback", rel=rel)
    assert np.isfinite(d["a_flat"]).all() and rel < 1e-3, rel          # (three steps: the fp32-atomic summation order of two runs, amplified by the steps)


def test_evaluate_model_two_ranks_write_what_one_process_writes(tmp_path):
    """Evaluation is "replicas only" (SURVEY.md 8e): under torchrun every rank of evaluate_model.py takes a strided shard of the image
    list and writes the WIDER result files of its images (the reference loop, evaluate_model.py:56-68, sharded).  Two ranks sharing
    the box's GPU (gloo rendezvous) must leave exactly the files a single process leaves, byte for byte."""
    import subprocess
    import sys
    PKG = os.path.join(ROOT, "tiny-faces-pytorch_amd")
    sys.path.insert(0, ROOT); sys.path.insert(0, PKG)
    from bench import tame_init_
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_objects=1, num_templates=25)
    tame_init_(m, seed=3)
    with torch.no_grad():
        for head in (m.score_res3, m.score_res4):
            head.bias[:25] -= 3.0
    ck = tmp_path / "tame.pth"
    torch.save({"epoch": 1, "batch_size": 4, "model": m.state_dict(), "optimizer": {}}, ck)
    evalm = os.path.join(PKG, "evaluate_model.py")
    common = ["synthetic", "--checkpoint", str(ck), "--num-images", "3", "--prob_thresh", "0.5"]
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r1 = subprocess.run([sys.executable, evalm] + common + ["--results_dir", str(tmp_path / "one")], cwd=tmp_path, capture_output=True, text=True,
                        timeout=400, env=env)
    assert r1.returncode == 0, r1.stdout[-1000:] + r1.stderr[-3000:]
    env2 = dict(env, TINYFACES_DIST_BACKEND="gloo", TINYFACES_SHARE_GPU="1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29547", evalm] + common + ["--results_dir", str(tmp_path / "two")], cwd=tmp_path, capture_output=True,
                        text=True, timeout=600, env=env2)
    assert r2.returncode == 0, r2.stdout[-1000:] + r2.stderr[-3000:]
    assert "evaluated on 2 ranks" in r2.stdout
    one, two = tmp_path / "one" / "synthetic", tmp_path / "two" / "synthetic"
    assert sorted(os.listdir(one)) == sorted(os.listdir(two)) == ["img_0.txt", "img_1.txt", "img_2.txt"]
    for f in os.listdir(one):
        assert open(one / f).read() == open(two / f).read(), f


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (the form the driver uses, no launcher around it) starts its N ranks itself.  On the
    1-GPU lease the two ranks share cuda:0 over gloo (TINYFACES_BENCH_SHARE_GPU=1; two RCCL ranks on one device are refused by RCCL):
    rc 0, ONE JSON line on stdout, n_gpus == 2, a whole-job value of two ranks' images and the exchange block of the N > 1 path."""
    import json
    env = dict(os.environ, TINYFACES_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["global_batch"] == 24 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["value"] > 0 and abs(out["value"] - 24 / (out["ms_per_step"] * 1e-3)) < 0.02 * out["value"]
    assert out["allreduce"].get("ranks") == 2, out["allreduce"]
    assert out["roofline"]["bound"] in ("hbm", "mfma") and 0 < out["roofline"]["frac"] < 1
    report("bench_self_launch", img_s=out["value"], ms_per_step=out["ms_per_step"], exposed_ms=out["allreduce"].get("exposed_ms_per_step"))
