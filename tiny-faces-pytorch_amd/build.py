#!/usr/bin/env python3
"""Build libtinyfaces_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python tiny-faces-pytorch_amd/build.py [--force]

Objects land in tiny-faces-pytorch_amd/build/, the library IN-TREE at
tiny-faces-pytorch_amd/tinyfaces/libtinyfaces_hip.so (git-ignored, travels with gpurun)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "tinyfaces", "libtinyfaces_hip.so")
EXPORTS = os.path.join(CSRC, "exports.map")   # only tf_* leaves the library
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# --experimental / TINYFACES_BUILD_EXPERIMENTAL=1: compile the measured-and-lost kernels in as well (csrc/common.h: TF_EXPERIMENTAL)
if os.environ.get("TINYFACES_BUILD_EXPERIMENTAL") or "--experimental" in sys.argv:
    COMMON = COMMON + ["-DTF_EXPERIMENTAL"]
# files whose float64 arithmetic must round exactly like numpy's: no FMA contraction
EXACT = {"targets.hip", "nms.hip", "decode.hip", "augment.hip"}


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "tinyfaces_hip.h"))
    hdrs.append(EXPORTS)
    return max(os.path.getmtime(h) for h in hdrs)


def source_digest():
    """tf_build_id(): sha256 over every source of the library (csrc/*, the public header) and the compiler flags, 16 hex digits."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(os.path.dirname(HERE), "include", "tinyfaces_hip.h")]
    for p in files:
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode() + b"\0")
            with open(p, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(COMMON + sorted(EXACT)).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_m = _deps_mtime()
    digest = source_digest()
    stamp = os.path.join(OBJ, "build_id.txt")
    restamp = not os.path.exists(stamp) or open(stamp).read().strip() != digest      # capi.o carries the digest: rebuilt whenever any source changed
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m) or (f == "capi.hip" and restamp):
            cmd = [HIPCC] + COMMON + (["-ffp-contract=off"] if f in EXACT else []) + ([f'-DTF_BUILD_ID="{digest}"'] if f == "capi.hip" else []) + ["-c", src, "-o", obj]
            jobs.append((f, cmd))
            if f == "conv_pws.hip":
                jobs.append(("conv_pws.hip [ISA audit]", None))

    def run(job):
        f, cmd = job
        if cmd is None:                                      # the ISA audit of conv_pws (isa_audit.py): a library whose tile loop breaks its ring rules is never linked
            import isa_audit
            try:
                bad = isa_audit.audit(isa_audit.emit_isa(HIPCC, os.path.join(CSRC, "conv_pws.hip"), os.path.join(OBJ, "conv_pws.s"), [x for x in COMMON if x.startswith("-D")]),
                                      min_kernels=30)
            except Exception as e:                           # noqa: BLE001
                return f, 1, str(e)
            return f, (1 if bad else 0), "\n".join(bad)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return f, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for f, rc, out in ex.map(run, jobs):
            if verbose:
                print(f"[hipcc] {f}: {'ok' if rc == 0 else 'FAILED'}")
            if rc != 0:
                raise RuntimeError(f"hipcc failed for {f}:\n{out}")
            if out.strip() and verbose:
                print(out)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if force or jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):     # (an object built by hand counts)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={EXPORTS}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[link] {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
