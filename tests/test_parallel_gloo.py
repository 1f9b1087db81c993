"""CPU, world_size 2, gloo: the data-parallel gradient path (tinyfaces/parallel.py)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tiny-faces-pytorch_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tinyfaces import parallel
    assert parallel.init_from_env("gloo")
    torch.manual_seed(rank)                                    # different init per rank on purpose
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    parallel.broadcast_module(lin)
    w0 = torch.cat([p.detach().reshape(-1) for p in lin.parameters()])
    # flat gradient buffer with adjacent views (what DetectionModel's backward produces) + one stray tensor
    params = list(lin.parameters())
    flat = torch.zeros(sum(p.numel() for p in params[:-1]))
    o = 0
    for p in params[:-1]:
        p.grad = flat[o:o + p.numel()].view_as(p)
        p.grad.fill_(float(rank + 1))
        o += p.numel()
    params[-1].grad = torch.full_like(params[-1], 10.0 * (rank + 1))
    red = parallel.GradientReducer(params, bucket_mb=1e-4)      # tiny buckets -> several all-reduces
    red.average_gradients()
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in params[:-1]) and \
        torch.allclose(params[-1].grad, torch.full_like(params[-1], 15.0))
    q.put((rank, w0.tolist(), bool(ok), bool(torch.allclose(flat, torch.full_like(flat, 1.5)))))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gradient_average_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] == res[1][1]                   # identical weights after broadcast
    assert all(r[2] and r[3] for r in res)          # averaged, and in place in the flat buffer


def _dying_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tiny-faces-pytorch_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      TINYFACES_DIST_TIMEOUT_S="20")
    import time
    from tinyfaces import parallel
    assert parallel.init_from_env("gloo")
    lin = torch.nn.Linear(6, 4)
    parallel.broadcast_module(lin)
    red = parallel.GradientReducer(list(lin.parameters()), bucket_mb=1e-4)
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    red.average_gradients()                                      # step 1: both ranks alive
    if rank == 1:
        os._exit(17)                                             # dies "mid-step": no barrier, no destroy, connections drop
    t0 = time.time()
    try:
        for p in lin.parameters():
            p.grad = torch.full_like(p, 1.0)
        red.average_gradients()                                  # step 2: the peer is gone
        q.put(("no error", time.time() - t0))
    except Exception as e:                                       # noqa: BLE001 (gloo raises RuntimeError / DistBackendError depending on the version)
        q.put((type(e).__name__, time.time() - t0))


def test_a_dead_rank_fails_the_survivor_fast_world2():
    """SURVEY.md section 5 (failure detection): rank 1 exits between two steps; rank 0's next gradient exchange must RAISE within the
    collective deadline (TINYFACES_DIST_TIMEOUT_S) instead of hanging or silently returning unreduced gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dying_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    what, dt = q.get(timeout=90)
    [p.join(30) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert what != "no error", "the survivor completed a collective with a dead peer"
    assert dt < 30.0, (what, dt)
    assert procs[1].exitcode == 17


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "tiny-faces-pytorch_amd"))
    from tinyfaces import parallel
    assert not parallel.is_distributed() and parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.reducer_for(torch.nn.Linear(2, 2)) is None
