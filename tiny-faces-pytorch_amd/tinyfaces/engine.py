"""Fused training engine: one reference training step (tinyfaces/trainer.py:72-87) with no autograd graph,
no per-parameter Python loop and no host sync:

    targets (HIP) -> forward (HIP executor) -> criterion fwd+bwd (HIP) -> backward (HIP executor)
    -> [RCCL all-reduce of the flat gradient, overlapped with the backward pass] -> fused SGD (one launch per group)

Data-parallel overlap: the flat gradient is cut along the backward order into buckets of ~`bucket_mb` MB (default 10: heads +
layer3.21-22 first, then THREE layer-3 bottlenecks of 4.46 MB each per bucket (two stay under 10 MB), ..., finally layer1/2 + stem: SURVEY.md 8e asks for
8-12 buckets of ~10 MB in layer 3).  When a bucket's gradients are enqueued the executor calls back with the stream that carries
them (tf_detnet_set_grad_callback, r3; rounds 1-2: an event + a communication stream waiting on it, still available) and that bucket's
all-reduce is issued there, while the remaining bottlenecks are still being differentiated: only the last, small bucket is exposed.

Semantics are those of main.py:67-70 (SGD momentum 0.9, weight decay 5e-4, the 4 learning-rate groups of
model.py:67-87) and of DetectionCriterion (loss.py).  `trainer.train` (autograd + torch.optim) remains the
drop-in path; this engine is what bench.py and the bundled main.py use."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import ops, parallel
from ._hip import lib


class TrainEngine:
    def __init__(self, model, criterion, lr=1e-4, momentum=0.9, weight_decay=5e-4, device="cuda", bucket_mb=10, native_exchange=None):
        self.device = torch.device(device)
        self.model = model.to(self.device).train()
        self.criterion = criterion
        if parallel.is_distributed():
            parallel.broadcast_module(self.model)
        self.flat_p = self.model.flatten_parameters()
        # the engine owns the flat storages from here on: the per-step identity walk over the 571 tensors is skipped (model._sync_tables;
        # anything that moves the module -- .to(), flatten_parameters -- unfreezes it).  TINYFACES_SYNC_TABLES_ALWAYS=1: the walk of rounds 1-3
        self.model._tables_frozen = os.environ.get("TINYFACES_SYNC_TABLES_ALWAYS") is None
        self.flat_m = torch.zeros_like(self.flat_p)
        self.groups = self.model.group_ranges()
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.steps = 0
        # r6: None = the default rule (_want_native): the library's own RCCL hook when the world is larger than one rank and librccl resolves,
        # the torch.distributed callback otherwise and whenever the native set-up or a native step reports an error
        self.native_exchange = native_exchange
        self.native_fallbacks = 0             # how often the engine left the native exchange (set-up refused / a bucket's collective failed)
        self.skip_allreduce = False          # measurement knob (bench.py): a step without the exchange, to size what the overlap hides
        self._overlap = None
        self._native = None
        # r3 experiment, NEGATIVE, opt-in (TINYFACES_SGD_PER_BUCKET=1): the gradient buckets (and the events the executor records when a
        # bucket is final) also drive the optimizer -- the SGD update of a bucket on the communication stream as soon as the bucket is
        # final (after its all-reduce when data-parallel), beside the rest of the backward pass, instead of over all 42.5 M parameters
        # at the end of the step.  Same numbers bit for bit; 1113-1117 img/s against 1118-1123 at the end (A/B on one box): the update
        # is HBM-bound (850 MB) and takes from the backward pass what it saves at the tail.
        self.sgd_per_bucket = bool(os.environ.get("TINYFACES_SGD_PER_BUCKET"))
        if parallel.is_distributed() or self.sgd_per_bucket:
            self._setup_overlap()

    # first bottleneck (executor index: layer1 0-2, layer2 3-6, layer3 7-29) of each bucket, in backward order: the coarse
    # 4-bucket cut of round 1, kept as an explicit choice (bucket_mb=0)
    _BUCKET_FIRST_BLOCK = (22, 14, 7)
    _BLOCK_NAMES = tuple(f"model.{l}.{i}." for l, n in (("layer1", 3), ("layer2", 4), ("layer3", 23)) for i in range(n))

    @classmethod
    def auto_first_blocks(cls, segments, total, bucket_mb):
        """First bottleneck of each bucket (backward order) so that every bucket but the last carries >= bucket_mb MB of fp32
        gradient: walk the bottlenecks from layer3.22 down, close a bucket as soon as it is big enough; layer 1/2 + stem (6 MB)
        always form the final bucket (event -1 = end of the backward pass)."""
        want = bucket_mb * (1 << 20) / 4
        firsts, end = [], total
        for b in range(len(cls._BLOCK_NAMES) - 1, 6, -1):             # layer 3 only: 29 .. 7
            start = min(o for k, (o, _) in segments.items() if k.startswith(cls._BLOCK_NAMES[b]))
            if end - start >= want:
                firsts.append(b)
                end = start
        if not firsts or firsts[-1] != 7:
            firsts.append(7)                                            # whatever is left of layer 3
        return tuple(firsts)

    @staticmethod
    def bucket_ranges(segments, total, first_blocks=_BUCKET_FIRST_BLOCK):
        """[(event block, start, end)] in backward order: bucket k covers the flat-gradient elements of every bottleneck
        >= first_blocks[k] not covered by an earlier bucket (plus the heads for k = 0); the last bucket (block -1 = "end
        of the backward pass") takes the rest (layer1, layer2, stem).  `segments` = DetectionModel._segments
        ({state_dict key: (offset, numel)} in executor order)."""
        names = [f"model.{l}.{i}." for l, n in (("layer1", 3), ("layer2", 4), ("layer3", 23)) for i in range(n)]
        ranges, end = [], total
        for b in first_blocks:
            start = min(o for k, (o, _) in segments.items() if k.startswith(names[b]))
            ranges.append((b, start, end))
            end = start
        ranges.append((-1, 0, end))
        return ranges

    def _setup_overlap(self):
        total = self.flat_p.numel()
        firsts = self.auto_first_blocks(self.model._segments, total, self.bucket_elems * 4 / (1 << 20)) if self.bucket_elems > 0 else self._BUCKET_FIRST_BLOCK
        ranges = [r for r in self.bucket_ranges(self.model._segments, total, firsts) if r[2] > r[1]]
        events = []
        for _ in ranges:
            ev = torch.cuda.Event()
            ev.record()                                   # materialises the hipEvent_t handle
            events.append(ev)
        blocks = (C.c_int * len(ranges))(*[r[0] for r in ranges])
        handles = (C.c_void_p * len(ranges))(*[int(ev.cuda_event) for ev in events])
        # the events belong to this engine's MODEL: DetectionModel._run_backward hands them to the executor for the duration of its
        # own backward call (r3: no process-wide registration any more, several engines per process are fine)
        self.model._grad_events = (blocks, handles, len(ranges))
        # r3: the all-reduce of a bucket is issued from INSIDE the backward call (tf_detnet_set_grad_callback), under the stream that
        # carries the bucket's last gradient kernel; the process group orders its own stream behind that point.  No communication stream
        # of ours any more (TINYFACES_ALLREDUCE_COMM_STREAM=1 brings the event-waiting stream of rounds 1-2 back).  Checked on one GPU with
        # a 1-rank RCCL group (TINYFACES_FORCE_DIST=1): 1106 img/s against 1124 without the collectives.
        self._use_comm_stream = bool(os.environ.get("TINYFACES_ALLREDUCE_COMM_STREAM")) or bool(os.environ.get("TINYFACES_ALLREDUCE_ON_MAIN"))
        # r4: the NATIVE exchange (TINYFACES_ALLREDUCE_NATIVE=1 / native_exchange=True): the executor's gradient hook is the library's own
        # tf_comm_allreduce_hook -- ncclAllReduce issued from C on the communicator's stream, no Python (no GIL) inside the backward enqueue,
        # the path a maintainer binding only the C ABI gets (include/tinyfaces_hip.h: tf_comm_*).  The communicator's identifier travels
        # through the torch.distributed group that already exists (any backend).  Opt-in: like the torch path it has only ever seen a
        # 1-rank group on hardware (tests/test_gpu_dist.py); the default stays the torch.distributed callback below.
        self._native = None
        if self._want_native() and not self._use_comm_stream and not self.sgd_per_bucket:
            self._native = self._setup_native(ranges)
        self._works, self._cb_error, self._ext_streams = [], None, {}
        self._block_range = {r[0]: (r[1], r[2]) for r in ranges}
        self._cb = C.CFUNCTYPE(None, C.c_int, C.c_void_p, C.c_void_p)(self._on_bucket)
        self.model._grad_callback = None if self._use_comm_stream else self._cb
        self.model._grad_callback_user = None
        if self._native is not None:          # the C hook + its plan instead of the ctypes callback
            self.model._grad_callback = C.cast(lib().tf_comm_allreduce_hook, C.c_void_p)
            self.model._grad_callback_user = C.cast(C.pointer(self._native["plan"]), C.c_void_p)
        self._overlap = dict(ranges=ranges, events=events, comm=torch.cuda.Stream(device=self.device) if self._use_comm_stream or self.sgd_per_bucket else None,
                             keep=(blocks, handles))

    def _want_native(self):
        """Which exchange (r6).  An explicit choice wins: `native_exchange=True / False`, TINYFACES_ALLREDUCE_NATIVE=1 / 0.
        Otherwise the C hook (ncclAllReduce issued by the library itself, no Python inside the backward enqueue) is the default as soon as there
        is more than one rank; `_setup_native` falls back to the torch.distributed callback when RCCL refuses (no librccl, ranks sharing a
        device, a failing ncclCommInitRank), and every rank takes the same decision."""
        env = os.environ.get("TINYFACES_ALLREDUCE_NATIVE")
        if self.native_exchange is not None:
            return bool(self.native_exchange)
        if env is not None:
            return env not in ("", "0")
        return parallel.is_distributed() and parallel.world_size() > 1

    def _all_ranks_ok(self, ok):
        """The same verdict on every rank: MIN over the torch.distributed group that carried us here (any backend)."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _setup_native(self, ranges):
        """The communicator + the hook's plan, or None (= use the torch.distributed callback) when any rank could not set it up."""
        from . import _hip
        explicit = self.native_exchange is True or os.environ.get("TINYFACES_ALLREDUCE_NATIVE") not in (None, "", "0")
        if not self._all_ranks_ok(bool(lib().tf_comm_available())):
            if explicit:
                raise RuntimeError("TINYFACES_ALLREDUCE_NATIVE: librccl.so could not be resolved at run time")
            self.native_fallbacks += 1
            return None
        ident = [bytes(_hip.TF_COMM_ID_BYTES)]
        if parallel.rank() == 0:
            buf = C.create_string_buffer(_hip.TF_COMM_ID_BYTES)
            if lib().tf_comm_unique_id(buf) == 0:
                ident = [buf.raw]
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast_object_list(ident, src=0)
        comm = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib().tf_comm_init(ident[0], parallel.rank(), parallel.world_size(), C.byref(comm)) if any(ident[0]) else -1
        if not self._all_ranks_ok(rc == 0):
            if rc == 0:
                lib().tf_comm_destroy(comm)
            if explicit:
                raise RuntimeError(f"native exchange: tf_comm_init failed on a rank (this rank: rc = {rc})")
            if parallel.rank() == 0:
                print("tinyfaces: the native RCCL exchange could not be set up (tf_comm_init); using the torch.distributed exchange", flush=True)
            self.native_fallbacks += 1
            return None
        n = len(ranges)
        blocks = (C.c_int * n)(*[r[0] for r in ranges])
        start = (C.c_int64 * n)(*[r[1] for r in ranges])
        end = (C.c_int64 * n)(*[r[2] for r in ranges])
        plan = _hip.CommPlan()
        plan.comm, plan.grad_flat, plan.n = comm, self.model._grad_flat_persistent.data_ptr(), n
        plan.blocks, plan.start, plan.end = C.cast(blocks, C.POINTER(C.c_int)), C.cast(start, C.POINTER(C.c_int64)), C.cast(end, C.POINTER(C.c_int64))
        plan.rc, plan.issued = 0, 0
        status = (C.c_int * n)()
        plan.status = C.cast(status, C.POINTER(C.c_int))
        return dict(comm=comm, plan=plan, keep=(blocks, start, end, status), n=n, status=status, ranges=ranges)

    def _on_bucket(self, block, stream_ptr, _user):
        """Called by the executor while it enqueues the backward pass: bucket `block` is final at the tail of `stream_ptr`."""
        try:
            if not (parallel.is_distributed() and not self.skip_allreduce) or self.sgd_per_bucket:
                return
            start, end = self._block_range[block]
            g = self.model._grad_flat_persistent
            cur = torch.cuda.current_stream(self.device)
            if (stream_ptr or 0) == cur.cuda_stream:
                self._works.append(dist.all_reduce(g[start:end], op=dist.ReduceOp.SUM, async_op=True))
                return
            ext = self._ext_streams.get(stream_ptr)
            if ext is None:
                ext = self._ext_streams[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=self.device)
            with torch.cuda.stream(ext):
                self._works.append(dist.all_reduce(g[start:end], op=dist.ReduceOp.SUM, async_op=True))
        except BaseException as e:       # an exception must not unwind through the C frames of the executor
            self._cb_error = e

    def _leave_native(self):
        """From the C hook back to the ctypes callback (same buckets, same events): after a native step that reported an error."""
        if self._native is not None:
            lib().tf_comm_destroy(self._native["comm"])
            self._native = None
        self.native_fallbacks += 1
        self.model._grad_callback = self._cb
        self.model._grad_callback_user = None

    def _drain_exchange(self):
        """Error path of step(): whatever the gradient hooks issued before the backward call failed is waited for, so that neither the
        communicator's stream nor torch's work objects outlive the step (a peer that did NOT fail is released by the group's timeout)."""
        try:
            if self._native is not None:
                with torch.cuda.device(self.device):
                    lib().tf_comm_join(self._native["comm"], torch.cuda.current_stream(self.device).cuda_stream)
                self._native["plan"].issued, self._native["plan"].rc = 0, 0
            works, self._works = getattr(self, "_works", []), []
            for w in works:
                w.wait()
        except Exception:
            pass

    def close(self):
        """Detach the gradient-ready events from the model (they are owned by this engine: the executor must not record
        handles that are about to be destroyed) and hand the pointer tables back to the per-call identity walk."""
        if getattr(self, "model", None) is not None:
            self.model._tables_frozen = False
        if self._overlap is not None:
            if getattr(self.model, "_grad_events", None) is not None and self.model._grad_events[0] is self._overlap["keep"][0]:
                self.model._grad_events = None
                self.model._grad_callback = None
                self.model._grad_callback_user = None
            if getattr(self, "_native", None) is not None:
                lib().tf_comm_destroy(self._native["comm"])
                self._native = None
            self._overlap = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lr(self, lr):
        self.lr = lr

    # ---- checkpoint interop with torch.optim.SGD (main.py:67-76,97-102) ------------------------------------------------
    def _group_params(self):
        """[(lr multiplier, [(state_dict key, nn.Parameter)])] in the order of DetectionModel.learnable_parameters
        (model.py:67-87), i.e. the parameter numbering torch.optim.SGD(model.learnable_parameters(lr)) uses."""
        m = self.model
        name_of = {id(p): k for k, p in m.named_parameters()}
        mults = (1.0, 0.1, 1.0, 0.0)
        return [(mult, [(name_of[id(p)], p) for p in g["params"]]) for mult, g in zip(mults, m.learnable_parameters(1.0))]

    def optimizer_state_dict(self, base_lr=None):
        """The engine's SGD state in the format of `torch.optim.SGD(model.learnable_parameters(lr), ...).state_dict()`, so that
        the reference's resume path (`optimizer.load_state_dict(checkpoint['optimizer'])`, main.py:76) accepts a checkpoint
        written by the fused trainer: one `momentum_buffer` per trained parameter (a slice of the flat momentum buffer), the
        four parameter groups with lr / initial_lr (StepLR needs it when constructed with last_epoch >= 0, main.py:81-83)."""
        base_lr = self.lr if base_lr is None else base_lr
        seg = self.model._segments
        state, groups, idx = {}, [], 0
        for mult, params in self._group_params():
            ids = []
            for k, p in params:
                if k in seg and mult != 0.0 and self.steps > 0:
                    o, n = seg[k]
                    state[idx] = {"momentum_buffer": self.flat_m[o:o + n].view_as(p).detach().clone().cpu()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": self.lr * mult, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay,
                           "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                           "initial_lr": base_lr * mult, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict: also accepts the state_dict of a real torch.optim.SGD built from
        model.learnable_parameters (the reference's own checkpoints, main.py:97-102).  Parameters without a
        momentum_buffer (never stepped, `model.fc.*`, lr-0 upsample) start from zero like torch does."""
        if not sd or not sd.get("param_groups"):
            return False
        seg = self.model._segments
        flat = [kp for _, params in self._group_params() for kp in params]
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(flat):
            raise ValueError(f"optimizer state has {len(ids)} parameters, the model has {len(flat)}")
        self.flat_m.zero_()
        loaded = 0
        for i, (k, p) in zip(ids, flat):
            st = sd["state"].get(i)
            buf = None if st is None else st.get("momentum_buffer")
            if buf is None or k not in seg:
                continue
            o, n = seg[k]
            if buf.numel() != n:
                raise ValueError(f"momentum_buffer of {k}: {buf.numel()} elements, expected {n}")
            self.flat_m[o:o + n].copy_(buf.reshape(-1).to(self.flat_m.device, torch.float32))
            loaded += 1
        g0 = sd["param_groups"][0]
        self.momentum = float(g0.get("momentum", self.momentum))
        self.weight_decay = float(g0.get("weight_decay", self.weight_decay))
        self.steps = max(self.steps, 1 if loaded else 0)
        return True

    def _allreduce(self, gflat):
        """Per bucket: the communication stream waits for the executor's gradient-ready event, then the all-reduce is
        issued from it (RCCL's own stream orders itself after the issuing stream); the compute stream only waits at
        the end.  Without the events (not set up): few large buckets after the whole backward pass."""
        if os.environ.get("TINYFACES_ALLREDUCE_ON_MAIN"):      # diagnostic: no communication stream, one synchronous collective per bucket behind the backward pass
            for (_, start, end) in (self._overlap["ranges"] if self._overlap is not None else [(0, 0, gflat.numel())]):
                dist.all_reduce(gflat[start:end], op=dist.ReduceOp.SUM)
            return
        if self._native is not None:
            # the collectives were issued by tf_comm_allreduce_hook during the backward call: the training stream waits for the communicator
            nat = self._native
            plan = nat["plan"]
            issued, rc = plan.issued, plan.rc
            status = list(nat["status"])
            plan.issued, plan.rc = 0, 0
            for k in range(nat["n"]):
                nat["status"][k] = 0
            from . import _hip
            with torch.cuda.device(self.device):
                _hip.check(lib().tf_comm_join(nat["comm"], torch.cuda.current_stream(self.device).cuda_stream), "tf_comm_join")
            if rc != 0 or issued != nat["n"]:
                # r6: a bucket whose collective could not be issued (or whose hook never ran) is reduced HERE through torch.distributed, behind the
                # join (the training stream has the whole gradient by now), and the engine leaves the native exchange for good: from the next
                # step on the ctypes callback issues every bucket.  (The hook fails on every rank alike or not at all -- an ncclAllReduce that one
                # rank cannot enqueue leaves its peers waiting in theirs, which only the group's timeout ends.)
                missing = [k for k in range(nat["n"]) if status[k] != 1]
                for k in missing:
                    _, start, end = nat["ranges"][k]
                    dist.all_reduce(gflat[start:end], op=dist.ReduceOp.SUM)
                if parallel.rank() == 0:
                    print(f"tinyfaces: native exchange: {len(missing)} of {nat['n']} gradient buckets were not issued (rc = {rc}); reduced through "
                          "torch.distributed, which carries the exchange from here on", flush=True)
                self._leave_native()
            return
        if self._overlap is not None and not self._use_comm_stream:
            # the collectives were issued by _on_bucket during the backward call: the training stream waits for them here
            works, self._works = self._works, []
            err, self._cb_error = self._cb_error, None
            if err is not None:
                raise err
            if len(works) != len(self._overlap["ranges"]):
                raise RuntimeError(f"data-parallel step: {len(works)} of {len(self._overlap['ranges'])} gradient buckets were reduced")
            for w in works:
                w.wait()
            return
        if self._overlap is not None:
            ov, works = self._overlap, []
            comm, cur = ov["comm"], torch.cuda.current_stream(self.device)
            sync_ops = not os.environ.get("TINYFACES_ALLREDUCE_ASYNC_OPS")
            for (_, start, end), ev in zip(ov["ranges"], ov["events"]):
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    # a SYNCHRONOUS collective issued under the communication stream: ProcessGroupNCCL (torch >= 2.8) launches it on the
                    # current stream, i.e. on `comm` itself, stream-ordered behind the event wait, and the host does not block
                    w = dist.all_reduce(gflat[start:end], op=dist.ReduceOp.SUM, async_op=not sync_ops)
                    if w is not None:
                        works.append(w)
            for w in works:
                w.wait()
            cur.wait_stream(comm)
            return
        works, end = [], gflat.numel()
        while end > 0:
            start = max(0, end - self.bucket_elems)
            works.append(dist.all_reduce(gflat[start:end], op=dist.ReduceOp.SUM, async_op=True))
            end = start
        for w in works:
            w.wait()

    @staticmethod
    def group_slices(groups, start, end):
        """[(a, b, lr multiplier)]: the parts of the flat range [start, end) that belong to a parameter group with a non-zero learning
        rate (model.py:67-87) -- a gradient bucket may straddle group boundaries (layer 3 | score_res3 | score_res4 | upsample)."""
        out = []
        for s, e, mult in groups:
            a, b = max(s, start), min(e, end)
            if mult != 0.0 and b > a:
                out.append((a, b, mult))
        return out

    def _bucket_updates(self, gflat, scale, reduce):
        """Per bucket, on the second stream: wait for the executor's gradient-ready event, all-reduce the bucket (data-parallel), then
        the SGD update of the bucket's parameters -- every (group ∩ bucket) range with the group's lr multiplier (model.py:67-87); the
        compute stream waits for that stream once, at the end of the step.  The masters of a bucket are only read again by the NEXT
        step's weight packing (the backward pass works on the packed copies and has left the bucket's layers when the event fires)."""
        ov = self._overlap
        comm, cur = ov["comm"], torch.cuda.current_stream(self.device)
        works = []
        for (_, start, end), ev in zip(ov["ranges"], ov["events"]):
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                works.append(dist.all_reduce(gflat[start:end], op=dist.ReduceOp.SUM, async_op=True) if reduce else None)
        for (_, start, end), w in zip(ov["ranges"], works):
            with torch.cuda.stream(comm):
                if w is not None:
                    w.wait()                              # the second stream (not the host, with RCCL) waits for this bucket's sum
                for a, b, mult in self.group_slices(self.groups, start, end):
                    ops.sgd_step(self.flat_p[a:b], gflat[a:b], self.flat_m[a:b], self.lr * mult, self.momentum, self.weight_decay, scale)
        cur.wait_stream(comm)

    def step(self, x, class_map, regression_map):
        """x (B,3,H,W) f32, class_map (B,nt,h,w) f32 (mined in place), regression_map (B,4nt,h,w) f32: all on the device.
        Returns the device tensor [sum cls loss, sum reg loss] (float64) without synchronising."""
        m, c = self.model, self.criterion
        m._sync_tables(x.device)
        out = m._run_forward(x, training=True)
        loss2, grad, _ = ops.criterion_fwd_bwd(out, class_map, regression_map, c.n_templates, c.reg_weight, c.ohem_thresh, c.max_pos,
                                               c.max_neg, c._pos_keep, c._neg_keep, c._next_seed())
        if self._native is not None:         # (bench.py measures a step without the exchange: the C hook is simply not installed for it)
            m._grad_callback = None if self.skip_allreduce else C.cast(lib().tf_comm_allreduce_hook, C.c_void_p)
            self._native["plan"].issued, self._native["plan"].rc = 0, 0       # a step that raised half-way must not poison the next one
        self._works, self._cb_error = [], None
        try:
            gflat = m._run_backward(x, grad, persistent=True)
        except BaseException:
            self._drain_exchange()               # collectives already queued by the hooks are joined before the error travels on
            raise
        scale = 1.0
        reduce = parallel.is_distributed() and not self.skip_allreduce
        if reduce:
            scale = 1.0 / parallel.world_size()          # average over ranks, folded into the SGD kernel
        if self.sgd_per_bucket and self._overlap is not None:
            self._bucket_updates(gflat, scale, reduce)
        else:
            if reduce:
                self._allreduce(gflat)
            for s, e, mult in self.groups:
                if mult == 0.0:
                    continue                              # score4_upsample: lr 0 (model.py:84) -> nothing to do
                ops.sgd_step(self.flat_p[s:e], gflat[s:e], self.flat_m[s:e], self.lr * mult, self.momentum, self.weight_decay, scale)
        self.steps += 1
        if hasattr(c, "_pending") and (parallel.rank() == 0 or not parallel.is_distributed()):
            c._pending.append((loss2, x.shape[0]))       # only the logging rank ever flushes the meters
        return loss2
