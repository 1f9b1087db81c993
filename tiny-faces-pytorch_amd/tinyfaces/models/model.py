"""DetectionModel -- same call surface as the reference's tinyfaces/models/model.py:7-128,
executed by the native HIP graph executor (csrc/detnet.hip) on MI355X.

What is kept from the reference:
  * constructor signature (model.py:12-16), `.forward(x)` (:89), `.learnable_parameters(lr)` (:67-87)
  * the exact state_dict (571 entries: torchvision resnet101 names under `model.` incl. the dead
    `model.fc.*` that model.py:23 leaves behind, `score_res3.*`, `score_res4.*`,
    `score4_upsample.weight` initialised to the bilinear kernel of model.py:45-65)
  * real nn.Parameters, so torch.optim.SGD / StepLR (main.py:67-83) work unchanged.
What is different: forward/backward never run torch ops.  One C-ABI call per pass launches the
hand-written gfx950 kernels; a CPU tensor raises (there is no CPU fallback).
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from .. import _hip
from .._hip import check, lib, ptr, stream

TF_DETNET_WEIGHTS_READY = 1      # include/tinyfaces_hip.h


class _Bottleneck(nn.Module):
    """Parameter container with torchvision's Bottleneck attribute names (never called)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class _ResNet101Trunk(nn.Module):
    """Parameter container with torchvision resnet101's names; `layer4` is deleted by the owner
    exactly like model.py:23, `avgpool`/`fc` stay (dead, but part of the checkpoint contract)."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 23, stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def resnet101(weights=None, **_):
    """Stand-in for `torchvision.models.resnet101` in the constructor signature (model.py:13)."""
    return _ResNet101Trunk()


class _DetNetFunction(torch.autograd.Function):
    """forward = tf_detnet_forward, backward = tf_detnet_backward (csrc/detnet.hip)."""

    @staticmethod
    def forward(ctx, x, owner, *params):
        out = owner._run_forward(x, training=True)
        ctx.owner = owner
        ctx.save_for_backward(x)
        ctx.ws_generation = owner._ws_generation
        return out

    @staticmethod
    def backward(ctx, gout):
        owner = ctx.owner
        (x,) = ctx.saved_tensors
        if ctx.ws_generation != owner._ws_generation:
            raise RuntimeError("DetectionModel: the activation arena was reused by a later forward before this "
                               "backward ran (one in-flight training step per model)")
        grads = owner._run_backward(x, gout.contiguous().float())
        return (None, None) + tuple(grads)


# Every register_parameter / register_buffer of ANY module in the process (Module.__setattr__ goes through them too) bumps this counter: a frozen
# pointer table (DetectionModel._sync_tables) is only trusted while nothing was registered since it was built.
_REGISTRATIONS = [0]


def _count_registration(module, name, value):
    _REGISTRATIONS[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)
torch.nn.modules.module.register_module_buffer_registration_hook(_count_registration)


def _invalidate_tables(module, incompatible_keys):
    """load_state_dict post-hook (a module-level function: the model stays picklable)."""
    module._table_key = None


class DetectionModel(nn.Module):
    """Hybrid-resolution Tiny Faces detector (model.py:7-128) on the MI355X executor."""

    def __init__(self, base_model=resnet101, pretrained_weights=None, num_templates=1, num_objects=1):
        super().__init__()
        output = (num_objects + 4) * num_templates                       # model.py:19
        self.num_out = output
        self.model = base_model(weights=None) if callable(base_model) else _ResNet101Trunk()
        if hasattr(self.model, "layer4"):
            del self.model.layer4                                        # model.py:23
        self.score_res3 = nn.Conv2d(512, output, 1)                      # model.py:25-28
        self.score_res4 = nn.Conv2d(1024, output, 1)                     # model.py:29-32
        self.score4_upsample = nn.ConvTranspose2d(output, output, 4, stride=2, padding=1, bias=False)   # :34-39
        self._init_bilinear()
        self.compute_dtype = _hip.tf_dtype(os.environ.get("TINYFACES_DTYPE", "bf16"))
        self._ws = None
        self._ws_generation = 0
        self._table_key = None
        self._frozen_tensors, self._frozen_version = (), -1
        self._session_depth = 0          # constant_weights() nesting
        self._ready_key = None           # (workspace ptr, dtype, table key) whose packed eval weights sit in the workspace
        self._lanes = []                 # forward_levels: extra (workspace, HIP stream, ready key) triples beside the model's own
        # load_state_dict(assign=True) re-points parameters without _apply: the pointer tables are rebuilt on the next call
        self.register_load_state_dict_post_hook(_invalidate_tables)
        if isinstance(pretrained_weights, (str, os.PathLike)):           # no network here: a local file only
            sd = torch.load(pretrained_weights, map_location="cpu", weights_only=True)
            sd = sd.get("model", sd)
            if any(k.startswith("model.") for k in sd):
                self.load_state_dict(sd, strict=False)
            else:                                                        # a bare torchvision resnet101 state_dict
                self.model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("layer4.")}, strict=False)
        elif pretrained_weights is not None:
            # the reference's default (ResNet101_Weights.IMAGENET1K_V1, model.py:13-14) is a download: impossible here.  Say so
            # loudly instead of silently training from random weights with the reference's schedule.
            import warnings
            warnings.warn(f"DetectionModel: pretrained_weights={pretrained_weights!r} is not a local file -- ignored, the trunk "
                          "keeps its RANDOM kaiming initialisation (no network to download ImageNet weights). Pass a path to a "
                          "torchvision resnet101 state_dict instead.", RuntimeWarning, stacklevel=2)

    # ---- reference surface ---------------------------------------------------------------
    def _init_bilinear(self):
        """model.py:45-65: diagonal bilinear kernel [.25,.75,.75,.25] (x) [.25,.75,.75,.25]."""
        k = self.score4_upsample.kernel_size[0]
        factor = np.floor((k + 1) / 2)
        center = factor if k % 2 == 1 else factor + 0.5
        c = np.arange(1, k + 1)
        v = np.ones((1, k)) - (np.abs(c - center) / factor)
        f = np.zeros((self.score4_upsample.in_channels, self.score4_upsample.out_channels, k, k))
        idx = np.arange(self.score4_upsample.out_channels)
        f[idx, idx] = v.T @ v
        self.score4_upsample.weight = nn.Parameter(torch.Tensor(f))

    def learnable_parameters(self, lr):
        """model.py:67-87: 4 SGD groups (trunk lr, score_res3 0.1*lr, score_res4 lr, upsample 0)."""
        return [{"params": self.model.parameters(), "lr": lr},
                {"params": self.score_res3.parameters(), "lr": 0.1 * lr},
                {"params": self.score_res4.parameters(), "lr": 1 * lr},
                {"params": self.score4_upsample.parameters(), "lr": 0}]

    def set_compute_dtype(self, dtype):
        """torch.float32 (exact-fp32 MFMA, parity path), torch.bfloat16 (fast path, training and inference) or torch.float16
        (fp16 MFMA operands, inference only: the hard-setting evaluation of BASELINE.json configs[4])."""
        self.compute_dtype = _hip.tf_dtype(dtype)
        return self

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("DetectionModel.forward: input is on the CPU. This build has no CPU path: the forward/"
                               "backward of the detector only exist as HIP kernels for MI355X (gfx950).")
        x = x.contiguous().float()
        self._sync_tables(x.device)
        if self.training and torch.is_grad_enabled():
            params = [p for p in self._grad_params]
            return _DetNetFunction.apply(x, self, *params)
        return self._run_forward(x, training=self.training)

    def forward_levels(self, xs, lanes=None):
        """`[self(x) for x in xs]` for the pyramid levels of ONE image (evaluation.py:49-68 runs them one after the other), eval mode only,
        with the levels spread over `lanes` HIP streams so that they run SIDE BY SIDE: the forward of a 1/4- or 1/16-size level is a chain
        of ~110 launches of 10-20 us each that fill a fraction of the 256 CUs (M = 1200-4800 pixels at layer 3: latency-bound whatever the
        kernel), and it hides completely beside the full-size level, which is throughput-bound.  Every lane owns a workspace (its packed
        weights + its activations: the executor is stateless beyond `ws`), the largest level runs on the caller's stream in the model's
        own workspace, the rest are dealt to the other lanes by decreasing size; the caller's stream waits for all lanes at the end.
        The outputs are those of the sequential loop bit for bit (same kernels, same launch geometry; tests/test_gpu_model.py).
        TINYFACES_EVAL_LANES (default 3) / lanes=1: the sequential loop."""
        xs = list(xs)
        n_lanes = int(os.environ.get("TINYFACES_EVAL_LANES", "3")) if lanes is None else int(lanes)
        if self.training or len(xs) < 2 or n_lanes < 2 or self._session_depth == 0:
            return [self(x) for x in xs]
        for x in xs:
            if not x.is_cuda:
                raise RuntimeError("DetectionModel.forward_levels: input is on the CPU (no CPU path)")
        xs = [x.contiguous().float() for x in xs]
        dev = xs[0].device
        self._sync_tables(dev)
        order, lane_of = self.assign_lanes([x.shape[0] * x.shape[2] * x.shape[3] for x in xs], n_lanes)
        while len(self._lanes) < n_lanes - 1:
            self._lanes.append({"ws": None, "stream": torch.cuda.Stream(device=dev, priority=int(os.environ.get("TINYFACES_EVAL_LANE_PRIO", "0"))), "ready": None})
        cur = torch.cuda.current_stream(dev)
        inputs_ready = cur.record_event()                                 # the level tensors were produced on the caller's stream
        outs = [None] * len(xs)
        used = set()
        big_first = os.environ.get("TINYFACES_EVAL_LANES_ORDER", "big") == "big"
        for i in (order if big_first else order[::-1]):                   # by decreasing size: the full-size level is enqueued first
            ln = lane_of[i]
            if ln == 0:
                outs[i] = self._run_forward(xs[i], training=False)
                continue
            lane = self._lanes[ln - 1]
            if ln not in used:
                lane["stream"].wait_event(inputs_ready)
                used.add(ln)
            outs[i] = self._run_eval_lane(xs[i], lane)
        for ln in used:
            cur.wait_stream(self._lanes[ln - 1]["stream"])
        return outs

    @staticmethod
    def assign_lanes(sizes, n_lanes):
        """(order, lane_of) for forward_levels: levels by decreasing size (ties: list order); the largest goes to lane 0 -- the caller's
        stream and the model's own workspace --, every further one to the lane with the least pixels so far (ties: lowest lane)."""
        order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
        load = [0] * n_lanes
        lane_of = {}
        for i in order:
            ln = 0 if not lane_of else min(range(n_lanes), key=lambda k: (load[k], k))
            lane_of[i] = ln
            load[ln] += sizes[i]
        return order, lane_of

    def _run_eval_lane(self, x, lane):
        """Eval-mode tf_detnet_forward of one level on a lane's own stream and workspace (forward_levels)."""
        N, _, H, W = x.shape
        H3, W3 = C.c_int(), C.c_int()
        lib().tf_detnet_out_shape(H, W, C.byref(H3), C.byref(W3))
        nbytes = lib().tf_detnet_workspace_bytes(self.compute_dtype, N, H, W, self.num_out, 0)
        if lane["ws"] is None or lane["ws"].device != x.device or lane["ws"].numel() < nbytes:
            lane["ws"] = None
            lane["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        ws = lane["ws"]
        out = torch.empty(N, self.num_out, H3.value, W3.value, dtype=torch.float32, device=x.device)
        key = (ws.data_ptr(), self.compute_dtype, self._table_key, self._session_serial)
        flags = TF_DETNET_WEIGHTS_READY if key == lane["ready"] else 0
        lane["ready"] = key
        bn = self.model.bn1
        with torch.cuda.device(x.device):
            check(lib().tf_detnet_forward_ctx(self._ctx(x.device), int(self.single_stream), self.compute_dtype, 0, ptr(x), N, H, W, self.num_out, self._param_ptrs,
                                              float(bn.eps), float(bn.momentum), ptr(out), ptr(ws), ws.numel(), flags, lane["stream"].cuda_stream),
                  "tf_detnet_forward_ctx")
        return out

    # ---- executor plumbing ---------------------------------------------------------------
    def _named_tensors(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def __getstate__(self):
        # pickling / copy.deepcopy: the executor's derived state (ctypes pointer tables, workspaces, streams) belongs to THIS object's storages;
        # a copy rebuilds it on its first call
        d = dict(self.__dict__)
        d.update(_table_key=None, _tables_frozen=False, _frozen_tensors=(), _frozen_version=-1, _ws=None, _ready_key=None, _lanes=[], _session_depth=0)
        for k in ("_param_ptrs", "_grad_params", "_bn_modules"):
            d.pop(k, None)
        return d

    def _apply(self, fn, *a, **kw):
        # .to() / .cuda() / .float(): every storage may move -> the pointer tables are rebuilt on the next call even when frozen
        self._table_key = None
        self._tables_frozen = False
        return super()._apply(fn, *a, **kw)

    def _sync_tables(self, device):
        """(Re)build the device-pointer tables the executor reads, keyed on storage identity.  The check itself walks 571 tensors
        (~0.4 ms of Python): an owner that pins the storages (TrainEngine after flatten_parameters) sets `_tables_frozen` and the walk is
        skipped until something moves the module (`_apply`, flatten_parameters)."""
        # (a constant_weights() session is the same promise as a frozen owner: the cheap check instead of the walk for every image of an evaluation loop)
        if (getattr(self, "_tables_frozen", False) or self._session_depth > 0) and self._table_key is not None:
            # frozen: no walk over the module tree, but EVERY tensor of the table is still checked (three sentinels missed a
            # replaced tensor in between).  Replacing a Parameter or a buffer anywhere registers it with its module, which bumps the global
            # registration counter below; re-pointing `.data` keeps the object and moves its storage, which the pointer comparison over the
            # cached tensor objects sees (571 data_ptr() calls, ~40 us); load_state_dict(assign=True) is caught by the post-hook
            if self._frozen_version == _REGISTRATIONS[0] and all(t.data_ptr() == p for t, p in zip(self._frozen_tensors, self._table_key)):
                return
        named = self._named_tensors()
        n = lib().tf_detnet_num_params()
        names = self._names if getattr(self, "_names", None) and len(self._names) == n else [lib().tf_detnet_param_name(i).decode() for i in range(n)]
        key = tuple(named[k].data_ptr() for k in names)
        if key == self._table_key:
            return
        for i, k in enumerate(names):
            t = named[k]
            if t.device != device:
                raise RuntimeError(f"parameter {k} is on {t.device}, input on {device}: call model.to(device) first")
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise RuntimeError(f"parameter {k} must be contiguous float32")
            want = lib().tf_detnet_param_numel(i, self.num_out)
            if t.numel() != want:
                raise RuntimeError(f"parameter {k}: {t.numel()} elements, executor expects {want}")
        up = self.score4_upsample.weight.detach()
        off = up.clone()
        idx = torch.arange(self.num_out, device=up.device)
        off[idx, idx] = 0
        if float(off.abs().sum()) != 0.0:
            raise RuntimeError("score4_upsample.weight has off-diagonal entries: the executor implements the frozen "
                               "per-channel bilinear upsample of model.py:45-65 (lr=0, model.py:84) only")
        self._names = names
        self._param_ptrs = (C.c_void_p * n)(*[named[k].data_ptr() for k in names])
        self._grad_names = [k for k in names if k in dict(self.named_parameters())]
        pd = dict(self.named_parameters())
        self._grad_params = [pd[k] for k in self._grad_names]
        self._grad_numels = [pd[k].numel() for k in self._grad_names]
        self._bn_modules = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        self._table_key = key
        self._frozen_tensors = [named[k] for k in names]
        self._frozen_version = _REGISTRATIONS[0]

    def _named_tensor(self, dotted):
        obj = self
        for part in dotted.split("."):
            obj = getattr(obj, part)
        return obj

    def flatten_parameters(self):
        """Re-point every parameter the executor trains into ONE flat fp32 buffer (executor order, each
        segment 16-byte aligned) and all BatchNorm counters into one int64 buffer.  The nn.Parameter objects
        keep their identity (optimizers stay valid); afterwards a whole parameter group is one contiguous
        range, so the fused SGD step and the RCCL all-reduce work on a handful of large segments."""
        dev = next(self.parameters()).device
        self._sync_tables(dev)
        pd = dict(self.named_parameters())
        total = sum((n + 3) // 4 * 4 for n in self._grad_numels)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._segments, o = {}, 0
        for k, n in zip(self._grad_names, self._grad_numels):
            p = pd[k]
            flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat[o:o + n].view_as(p)
            self._segments[k] = (o, n)
            o += (n + 3) // 4 * 4
        self._flat_params = flat
        nbt = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        cnt = torch.zeros(len(nbt), dtype=torch.int64, device=dev)
        for i, m in enumerate(nbt):
            cnt[i] = m.num_batches_tracked
            m.num_batches_tracked = cnt[i]
            m._buffers["num_batches_tracked"] = cnt[i:i + 1].view(())
        self._flat_nbt = cnt
        self._grad_flat_persistent = torch.zeros(total, dtype=torch.float32, device=dev)
        self._table_key = None
        self._sync_tables(dev)
        return flat

    def group_ranges(self):
        """[(start, end, lr_multiplier)] of the 4 reference parameter groups (model.py:67-87) inside the flat buffer."""
        seg = self._segments
        def span(prefix):
            ks = [k for k in self._grad_names if k.startswith(prefix)]
            return seg[ks[0]][0], seg[ks[-1]][0] + (seg[ks[-1]][1] + 3) // 4 * 4
        return [span("model.") + (1.0,), span("score_res3.") + (0.1,), span("score_res4.") + (1.0,), span("score4_upsample.") + (0.0,)]

    def _ctx(self, device):
        """The executor context of this model on `device` (include/tinyfaces_hip.h: tf_detnet_ctx -- second stream + event pool): created on
        first use with that device current, destroyed with the model.  Two models never share a stream or an event."""
        ctxs = self.__dict__.setdefault("_exec_ctx", {})
        key = (device.type, device.index)
        if key not in ctxs:
            h = C.c_void_p()
            with torch.cuda.device(device):
                check(lib().tf_detnet_ctx_create(C.byref(h)), "tf_detnet_ctx_create")
            ctxs[key] = h
        return ctxs[key]

    def __del__(self):
        for h in self.__dict__.get("_exec_ctx", {}).values():
            try:
                lib().tf_detnet_ctx_destroy(h)
            except Exception:
                pass

    single_stream = False            # True: weight gradients on the caller's stream (A/B + race-screen tests)

    def _workspace(self, device, nbytes):
        if self._ws is None or self._ws.device != device or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    @contextlib.contextmanager
    def constant_weights(self, reserve=None):
        """Promise that no parameter/buffer value changes inside the block (eval mode): the executor packs the weights
        and folds the BN statistics on the first forward and reuses them for every later one, whatever the image size.
        The pyramid of get_detections (evaluation.py:49-82) is 3-5 forwards on constant weights.  `reserve=(N, H, W)` sizes
        the workspace for the largest input up front so that it is not re-allocated (which would drop the packed copy)."""
        if reserve is not None:
            dev = next(self.parameters()).device
            self._workspace(dev, lib().tf_detnet_workspace_bytes(self.compute_dtype, *reserve, self.num_out, 0))
        if self._session_depth == 0:
            self._ready_key = None
            self._session_serial = getattr(self, "_session_serial", 0) + 1     # the lanes of forward_levels re-pack once per session too
        self._session_depth += 1
        try:
            yield self
        finally:
            self._session_depth -= 1
            if self._session_depth == 0:
                self._ready_key = None

    def _run_forward(self, x, training):
        N, _, H, W = x.shape
        H3, W3 = C.c_int(), C.c_int()
        lib().tf_detnet_out_shape(H, W, C.byref(H3), C.byref(W3))
        nbytes = lib().tf_detnet_workspace_bytes(self.compute_dtype, N, H, W, self.num_out, int(training))
        ws = self._workspace(x.device, nbytes)
        self._ws_generation += 1
        self._ws_shape = (N, H, W)
        out = torch.empty(N, self.num_out, H3.value, W3.value, dtype=torch.float32, device=x.device)
        bn = self.model.bn1
        flags = 0
        if training:
            self._ready_key = None
        elif self._session_depth > 0:
            key = (ws.data_ptr(), self.compute_dtype, self._table_key)
            if key == self._ready_key:
                flags = TF_DETNET_WEIGHTS_READY
            self._ready_key = key
        with torch.cuda.device(x.device):
            check(lib().tf_detnet_forward_ctx(self._ctx(x.device), int(self.single_stream), self.compute_dtype, int(training), ptr(x), N, H, W, self.num_out,
                                              self._param_ptrs, float(bn.eps), float(bn.momentum), ptr(out), ptr(ws), ws.numel(), flags, stream()),
                  "tf_detnet_forward_ctx")
        if training:
            if getattr(self, "_flat_nbt", None) is not None:
                self._flat_nbt += 1                                      # all 94 counters in one launch
            else:
                for m in self._bn_modules:                               # BatchNorm2d bookkeeping
                    if m.num_batches_tracked is not None:
                        m.num_batches_tracked += 1
        return out

    def _run_backward(self, x, gout, persistent=False):
        """persistent=False (autograd path): a fresh flat gradient buffer per call, returned as per-parameter views.
        persistent=True (fused trainer): gradients land in self._grad_flat_persistent (same layout as the flat
        parameter buffer); returns that buffer."""
        N, _, H, W = x.shape
        assert (N, H, W) == self._ws_shape
        total = sum((n + 3) // 4 * 4 for n in self._grad_numels)       # every segment 16-byte aligned
        if persistent:
            gflat = self._grad_flat_persistent
            cache = getattr(self, "_persist_table", None)
        else:
            gflat = torch.zeros(total, dtype=torch.float32, device=x.device)
            cache = None
        if cache is None or cache[0] != gflat.data_ptr():
            ptrs, o = {}, 0
            for k, n in zip(self._grad_names, self._grad_numels):
                ptrs[k] = gflat.data_ptr() + 4 * o
                o += (n + 3) // 4 * 4
            table = (C.c_void_p * len(self._names))(*[ptrs.get(k, 0) for k in self._names])
            if persistent:
                self._persist_table = (gflat.data_ptr(), table)
        else:
            table = cache[1]
        # gradient-ready hooks of a data-parallel TrainEngine live in THIS model (`_grad_events`, `_grad_callback`, set by the engine that
        # owns it) and travel as an ARGUMENT of this model's own backward call (tf_detnet_hooks, r4): nothing is registered process-wide,
        # several models / engines / threads never see each other's events.  The callback reduces slices of the PERSISTENT flat buffer, so
        # the autograd path (fresh buffer per call; trainer.train reduces behind the backward pass, parallel.GradientReducer) never gets it.
        ev = getattr(self, "_grad_events", None)
        cb = getattr(self, "_grad_callback", None) if (ev is not None and persistent) else None
        hooks = _hip.DetnetHooks()
        if ev is not None:
            hooks.blocks, hooks.events, hooks.n = C.cast(ev[0], C.POINTER(C.c_int)), C.cast(ev[1], C.POINTER(C.c_void_p)), ev[2]
        hooks.fn = C.cast(cb, C.c_void_p) if cb is not None else None
        hooks.user = getattr(self, "_grad_callback_user", None) if cb is not None else None
        hooks.single_stream = int(self.single_stream)
        with torch.cuda.device(x.device):
            check(lib().tf_detnet_backward_ctx(self._ctx(x.device), C.byref(hooks), self.compute_dtype, ptr(x), N, H, W, self.num_out, self._param_ptrs, table,
                                               ptr(gout), ptr(gflat), gflat.numel() * 4, ptr(self._ws), self._ws.numel(), stream()), "tf_detnet_backward_ctx")
        self._last_grad_flat = gflat
        if persistent:
            return gflat
        pd = dict(self.named_parameters())
        views, o = [], 0
        for k, n in zip(self._grad_names, self._grad_numels):
            views.append(gflat[o:o + n].view_as(pd[k]))
            o += (n + 3) // 4 * 4
        return views
