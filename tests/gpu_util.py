"""Helpers shared by the -m gpu parity tests: a report file the GPU box ships back
(gpurun_out/parity_report.txt) so one gpurun call yields numbers for every kernel."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def report(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    clean = {k: (float(v) if isinstance(v, (np.floating, float)) else (int(v) if isinstance(v, (np.integer, int)) else str(v)))
             for k, v in kw.items()}
    with open(os.path.join(OUT, "parity_report.txt"), "a") as f:
        f.write(json.dumps({"test": name, **clean}) + "\n")


def err(a, b):
    """max abs error, max |b|, relative-to-max error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = float(np.abs(a - b).max()) if a.size else 0.0
    m = float(np.abs(b).max()) if b.size else 0.0
    return d, m, d / (m + 1e-30)


def to_nhwc(x_nchw, dtype, device="cuda"):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(device=device, dtype=dtype)


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, dtype):
    """Round to the compute dtype and back (so the torch reference sees the same operand values)."""
    return x.to(dtype).float()
