"""Data-parallel training: one process per GPU, gradient all-reduce (average) over RCCL / xGMI.

The reference has no distributed code at all; this is the new 8-GPU path.
Semantics: per-rank mini-batches of 12, per-device BatchNorm statistics (the reference has no
SyncBN), gradients AVERAGED over ranks (the loss is a per-rank SUM, tinyfaces/models/loss.py:87-88,
so averaging keeps the single-GPU step size), identical initial weights via broadcast from rank 0.

xGMI is a point-to-point full mesh (7 links per GPU): the whole gradient (27.7 M fp32 = 111 MB)
is reduced as a few large flat buckets so RCCL can spread each one over all links; buckets are
views of the flat gradient buffer DetectionModel's backward already produces, so there is no
flatten/unflatten copy.  Buckets are launched async in reverse-execution order (heads first)."""
import os

import torch
import torch.distributed as dist


_FORCE = os.environ.get("TINYFACES_FORCE_DIST") == "1"      # test knob: run the collective path with a 1-rank group


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    backend 'nccl' is RCCL on ROCm; 'gloo' for the CPU tests."""
    if _FORCE and "RANK" not in os.environ:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    elif "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        # failure detection: a rank that dies mid-step must take the job down, not hang it -- every collective gets a
        # deadline (TINYFACES_DIST_TIMEOUT_S, default 600 s; gloo additionally sees the closed connection at once), and with RCCL the
        # watchdog aborts the communicator when it expires (TORCH_NCCL_ASYNC_ERROR_HANDLING defaults to 1 in torch >= 2.2)
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("TINYFACES_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, timeout=timeout)
    return True


def broadcast_module(module, src=0):
    """Same weights and buffers on every rank (one flat broadcast per dtype)."""
    if not is_distributed():
        return
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        o = 0
        for t in ts:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()


class GradientReducer:
    """Averages .grad of the given parameters across ranks with few large all-reduces."""

    def __init__(self, params, bucket_mb=32):
        self.params = [p for p in params]
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)

    def _buckets(self, grads):
        """Group grads (reverse order = order in which backward produced them) into flat views when
        they are adjacent slices of one storage (DetectionModel's flat gradient), else copy-flatten."""
        out, cur, cur_n = [], [], 0
        for g in reversed(grads):
            cur.append(g)
            cur_n += g.numel()
            if cur_n >= self.bucket_elems:
                out.append(cur)
                cur, cur_n = [], 0
        if cur:
            out.append(cur)
        return out

    @staticmethod
    def _as_flat_view(gs):
        base = min(gs, key=lambda g: g.data_ptr())
        total = sum(g.numel() for g in gs)
        lo, hi = base.data_ptr(), max(g.data_ptr() + g.numel() * 4 for g in gs)
        if hi - lo == total * 4 and all(g.is_contiguous() and g.dtype == torch.float32 for g in gs) \
                and base.untyped_storage().data_ptr() == gs[0].untyped_storage().data_ptr():
            try:
                return torch.as_strided(base, (total,), (1,), base.storage_offset())
            except RuntimeError:
                return None
        return None

    def average_gradients(self):
        if not is_distributed():
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        ws = dist.get_world_size()
        works = []
        for gs in self._buckets(grads):
            flat = self._as_flat_view(gs)
            copied = flat is None
            if copied:
                flat = torch.cat([g.reshape(-1) for g in gs])
            works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, gs, copied))
        for work, flat, gs, copied in works:
            work.wait()
            flat.div_(ws)
            if copied:
                o = 0
                for g in gs:
                    g.copy_(flat[o:o + g.numel()].view_as(g))
                    o += g.numel()


def reducer_for(model):
    if not is_distributed():
        return None
    r = getattr(model, "_tf_reducer", None)
    if r is None:
        broadcast_module(model)
        r = GradientReducer([p for p in model.parameters() if p.requires_grad])
        model._tf_reducer = r
    return r
