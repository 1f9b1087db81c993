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


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "tiny-faces-pytorch_amd"))
    from tinyfaces import parallel
    assert not parallel.is_distributed() and parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.reducer_for(torch.nn.Linear(2, 2)) is None
