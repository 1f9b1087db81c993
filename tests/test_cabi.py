"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/tinyfaces_hip.h
declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import importlib.util
    spec = importlib.util.spec_from_file_location("tf_build", os.path.join(ROOT, "tiny-faces-pytorch_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def _declared(experimental=False):
    src = open(os.path.join(ROOT, "include", "tinyfaces_hip.h")).read()
    if not experimental:       # the default library does not carry the `#ifdef TF_EXPERIMENTAL` entry points (measured-and-lost kernels)
        src = re.sub(r"#ifdef TF_EXPERIMENTAL.*?#endif /\* TF_EXPERIMENTAL \*/", "", src, flags=re.S)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built, hip):
    import ctypes
    l = ctypes.CDLL(built)
    declared = _declared(hip.experimental())
    assert len(declared) >= 35
    if not hip.experimental():      # r6: the default library is WITHOUT the kernels that lost (conv_pwx: tf_conv2d_bnbwd / tf_conv2d_bnfwd)
        assert not hasattr(l, "tf_conv2d_bnfwd") and not hasattr(l, "tf_conv2d_bnbwd")
        assert set(_declared(True)) - set(declared) == {"tf_conv2d_bnbwd", "tf_conv2d_bnfwd"}
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/tinyfaces_hip.h but not exported"
    listed = hip.symbols()
    assert set(declared) == set(listed)
    assert hip.lib().tf_version() >= 100


def test_library_exports_only_the_c_abi(built):
    """csrc/exports.map: nothing but tf_* (the ABI + the four debug probes of csrc/debug_api.h) is a dynamic symbol of the library --
    no C++-mangled launcher leaks out."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert names and not [n for n in names if n.startswith("_Z")], [n for n in names if n.startswith("_Z")][:5]
    dbg = set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "tiny-faces-pytorch_amd", "csrc", "debug_api.h")).read()))
    import ctypes
    fn = ctypes.CDLL(built).tf_build_id
    fn.restype = ctypes.c_char_p
    exp = fn().decode().endswith("+x")
    assert set(names) == set(_declared(exp)) | dbg


def test_binding_signatures_cover_the_abi(hip):
    assert set(hip._SIGNATURES) == set(hip.symbols())


def test_executor_parameter_table_matches_state_dict(hip):
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    sd = m.state_dict()
    l = hip.lib()
    n = l.tf_detnet_num_params()
    names = [l.tf_detnet_param_name(i).decode() for i in range(n)]
    assert len(set(names)) == n == 475
    for i, k in enumerate(names):
        assert k in sd, k
        assert sd[k].numel() == l.tf_detnet_param_numel(i, 125), k
    # everything in the state_dict that the executor does not read is dead weight of the reference (D4) or bookkeeping
    rest = [k for k in sd if k not in names]
    assert all(k.startswith("model.fc.") or k.endswith("num_batches_tracked") for k in rest)


def test_workspace_and_shape_queries(hip):
    import ctypes as C
    l = hip.lib()
    h3, w3 = C.c_int(), C.c_int()
    for (h, w), exp in {(500, 500): (63, 63), (501, 503): (63, 63), (480, 640): (60, 80), (960, 1280): (120, 160),
                        (64, 64): (8, 8), (1000, 1000): (125, 125)}.items():      # SURVEY.md section 3.3 [probe]
        l.tf_detnet_out_shape(h, w, C.byref(h3), C.byref(w3))
        assert (h3.value, w3.value) == exp
    train = l.tf_detnet_workspace_bytes(hip.TF_BF16, 12, 500, 500, 125, 1)
    evalb = l.tf_detnet_workspace_bytes(hip.TF_BF16, 12, 500, 500, 125, 0)
    assert 1e9 < evalb < train < 2e10
    assert l.tf_nms_workspace_bytes(4096) > 4096 * 64 * 8
    assert l.tf_criterion_workspace_bytes(12, 25, 63, 63) > 0 and l.tf_targets_workspace_bytes(100) >= 1200


def test_kernel_selection_queries_host_side(hip):
    """The host-side planning functions that decide which MFMA kernel a call gets (no device work): the statistic-row count of a
    conv launch (tf_conv_mtiles: the row count its BN consumer must read) and the partial-tile workspace of a weight gradient
    (tf_wgrad_workspace_bytes), for the shapes of the training step at bs = 12."""
    import ctypes as C
    l = hip.lib()
    rows = l.tf_get_stat_rows()
    assert 1 <= rows <= 16

    def conv(N, H, W, Cin, Cout, K, stride=1, dtype=hip.TF_BF16, tile=0):
        a = hip.ConvArgs()
        a.dtype, a.mode, a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = dtype, 0, N, H, W, Cin, Cout, K, K, stride, K // 2
        a.OH, a.OW = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
        a.ldy, a.tile = Cout, tile
        return l.tf_conv_mtiles(C.byref(a))

    # layer 3 at bs = 12 (12 x 32 x 32 pixels): the 3x3 goes to the halo-resident kernel (96 tiles of 4 x 32 pixels), the 1x1s to the
    # LDS-DMA kernel (192 / 96 tiles): all fold their partial sums into the same <= TF_STAT_ROWS rows
    assert conv(12, 32, 32, 256, 256, 3) == rows and conv(12, 32, 32, 1024, 256, 1) == rows and conv(12, 32, 32, 256, 1024, 1) == rows
    # few tiles: one row per tile.  4 x 32 output tiles of conv3x3h when forced (tile 50): 1 image x 2 x 1 tiles
    assert conv(1, 8, 32, 256, 256, 3, tile=50) == 2
    assert conv(1, 8, 8, 64, 64, 1) == 1                         # 64 pixels: one 64 x 64 / 128 x 64 tile
    # fp32 (the parity path) never takes the 2-byte kernels
    assert conv(1, 16, 16, 256, 256, 3, dtype=hip.TF_F32) == min(rows, 4)      # 256 pixels / 64
    # conv_pws (tile 70): the layer-1 launches take it (folded rows); an explicit 70 on a launch it does not take is refused here as in tf_conv2d,
    # never answered with a row count the caller would size a buffer from
    assert conv(12, 125, 125, 64, 256, 1) == rows and conv(12, 125, 125, 64, 256, 1, tile=70) == rows
    assert conv(12, 32, 32, 1024, 256, 1, tile=70) == -3 and conv(1, 8, 8, 64, 64, 1, tile=70) == -3      # TF_ERR_UNSUPPORTED: Cin 1024; 64 pixels

    def wgrad_ws(N, H, W, Cin, Cout, K, stride=1):
        a = hip.WgradArgs()
        a.dtype, a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW, a.stride, a.pad = hip.TF_BF16, N, H, W, Cin, Cout, K, K, stride, K // 2
        a.OH, a.OW = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
        a.ldx, a.lddy, a.dw_ld = Cin, Cout, Cin * K * K
        return l.tf_wgrad_workspace_bytes(C.byref(a))

    tile_bytes = 9 * 64 * 64 * 4
    # layer-3 3x3: 16 tiles of 64 x 64 x 9 taps, 8 pixel slices (128 blocks, r6): 8 x 16 partial tiles
    assert wgrad_ws(12, 32, 32, 256, 256, 3) == 8 * 16 * tile_bytes
    # layer-1 3x3 (one tile): at most 128 slices (r6), every slice >= 6 stages of 64 padded pixels
    w1 = wgrad_ws(12, 125, 125, 64, 64, 3)
    assert w1 % tile_bytes == 0 and 64 <= w1 // tile_bytes <= 128
    assert w1 <= (256 + 16) * tile_bytes                          # fits the executor's scratch (csrc/detnet.hip: P.dwp_floats)
    assert wgrad_ws(12, 32, 32, 256, 1024, 1) == 0                # pointwise: atomics, no workspace
    assert wgrad_ws(12, 63, 63, 128, 128, 3, stride=2) == 0       # strided 3x3: the per-tap kernel


def test_conv_pwx_k_loop_holds_only_the_hand_counted_waits(tmp_path):
    """csrc/conv_pwx.hip (r5 rewrite) streams its raw pixel stages through a 4-10 deep LDS-DMA ring issued by waves 4-7 and transforms them in
    place.  Two things hipcc does on its own would flatten that ring to one stage, and both were seen in the ISA of the first build: a
    `s_waitcnt vmcnt(0)` in front of the first ds_write behind a DMA it knows about (the builtin is modelled as a pending LDS write), and
    `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of every __syncthreads().  ISA audit of every instantiation: every DMA is issued from inline
    asm, and in the basic blocks of the K loop every wait on the VM counter is a hand-written one; no scratch."""
    import re
    import subprocess
    src = os.path.join(ROOT, "tiny-faces-pytorch_amd", "csrc", "conv_pwx.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DTF_EXPERIMENTAL", "-save-temps", "-c", src, "-o", str(tmp_path / "pwx.o")],      # (r6: an experimental-build kernel)
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(tmp_path / "conv_pwx-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    names = re.findall(r"^(_ZN12_GLOBAL__N_115conv_pwx_kernel\w+):", asm, re.M)
    assert len(names) >= 10           # {128, 256} x {plain, bn-bwd (2 epilogue forms), bn-fwd identity / downsample (2 each)}
    for name in names:
        body = re.search(re.escape(name) + r":(.*?)\.Lfunc_end", asm, re.S).group(1)
        assert "scratch_" not in body, name
        lines = [l.strip() for l in body.split("\n")]
        in_asm, hand = False, set()
        for i, l in enumerate(lines):
            if l.startswith(";;#ASMSTART"):
                in_asm = True
            elif l.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm:
                hand.add(i)
        dmas = [i for i, l in enumerate(lines) if l.startswith("global_load_lds")]
        assert dmas and all(i in hand for i in dmas), name
        # basic blocks of the K loop = the blocks hipcc annotates with the loop header of the block that holds the MFMAs
        owner, cur = {}, None
        for i, l in enumerate(lines):
            m = re.match(r"\.L(BB\d+_\d+):\s*;(.*)", l)
            if m:
                h = re.search(r"Header=(BB\d+_\d+)", m.group(2))
                cur = h.group(1) if h else (m.group(1) if "Loop Header" in m.group(2) else None)
            elif re.match(r"\.L(BB\d+_\d+):", l):
                cur = None
            owner[i] = cur
        kloops = {owner[i] for i, l in enumerate(lines) if l.startswith("v_mfma")}
        assert len(kloops) == 1 and None not in kloops, (name, kloops)
        loop = [i for i in range(len(lines)) if owner[i] in kloops]
        waits = [i for i in loop if lines[i].startswith("s_waitcnt") and "vmcnt" in lines[i]]
        assert waits, name
        assert all(i in hand for i in waits), (name, [lines[i] for i in waits if i not in hand])


def _barriers_with_lds_in_flight(asm):
    """{kernel: [(lds ops in flight at the barrier, a DMA is issued behind it before the next barrier)]} from hipcc's gfx950 assembly: a
    linear walk that counts ds_* instructions up and applies every `s_waitcnt lgkmcnt(n)`."""
    import re
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        lines = [l.strip() for l in m.group(2).splitlines()]
        if not any(l.startswith("global_load_lds") for l in lines):
            continue
        pend, hits = 0, []
        for i, l in enumerate(lines):
            if re.match(r"ds_(read|write|load|store)", l):
                pend += 1
            elif l.startswith("s_waitcnt"):
                w = re.search(r"lgkmcnt\((\d+)\)", l)
                if w:
                    pend = min(pend, int(w.group(1)))
            elif l.startswith("s_barrier") and pend:
                dma = False
                for k in range(i + 1, min(i + 60, len(lines))):
                    if lines[k].startswith("s_barrier"):
                        break
                    if lines[k].startswith("global_load_lds"):
                        dma = True
                        break
                hits.append((pend, dma))
        out[m.group(1)] = hits
    return out


@pytest.mark.parametrize("src", ["conv3x3h.hip", "conv_dma_bf16.hip", "wgrad_dma.hip", "wgrad3x3.hip", "conv_pwx.hip"])
def test_no_lds_read_is_in_flight_across_a_barrier_that_frees_its_ring_slot(tmp_path, src):
    """r5, the race behind `get_detections` returning different boxes from run to run (csrc/conv3x3h.hip): hipcc sank the last MFMAs of a K
    stage, and the wait for the fragment reads that feed them, below the NEXT stage's counted vmcnt wait + raw s_barrier; a wave then
    passed the barrier with reads of a ring slot still in flight, and the first thing a wave does behind that barrier is issue the LDS-DMA
    that refills that very slot.  Harmless until three pyramid levels shared the GPU.  The fix drains lgkmcnt in the stage wait; this
    audit keeps every ring kernel honest: in hipcc's assembly no barrier that is followed by a DMA issue may be reached with an LDS access
    outstanding (the debugging instantiation of conv3x3h with its time stamps is exempt; the grouped weight gradient reads AHEAD from a
    slot that is not the one being refilled and is not audited here)."""
    import subprocess
    path = os.path.join(ROOT, "tiny-faces-pytorch_amd", "csrc", src)
    out = tmp_path / "k.s"
    exp = ["-DTF_EXPERIMENTAL"] if src == "conv_pwx.hip" else []        # (r6: conv_pwx is an experimental-build kernel; the others are audited as shipped)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + exp + ["-S", "--cuda-device-only", path, "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found = _barriers_with_lds_in_flight(open(out).read())
    assert found                                         # the file does hold LDS-DMA kernels
    bad = {k: [h for h in v if h[1]] for k, v in found.items() if "Lb1E" not in k or "conv3x3h" not in k}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, bad


def test_conv_pws_tile_loop_keeps_its_ring_rules(tmp_path):
    """csrc/conv_pws.hip (r5): every wave walks its own 16-pixel tiles through a private two-slot LDS ring with NO block barrier in the loop, so
    the two orderings a barrier would have given are kept by wait counts alone.  The ISA audit (tiny-faces-pytorch_amd/isa_audit.py, also run by
    build.py whenever the file is compiled: ADVICE r5) over every instantiation: no scratch; every DMA is inline asm; in the tile loop every wait
    on the VM counter is the hand-counted one and its count equals the stores + DMAs the loop body really holds; no DMA of the loop is issued with
    an LDS access of this wave outstanding.  And the audit itself can fail: a loop with one store fewer is reported."""
    import re
    import isa_audit
    src = os.path.join(ROOT, "tiny-faces-pytorch_amd", "csrc", "conv_pws.hip")
    asm = isa_audit.emit_isa("/opt/rocm/bin/hipcc", src, str(tmp_path / "pws.s"))
    assert isa_audit.audit(asm) == []
    # negative control: drop ONE output store from the first kernel's tile loop -> the count check must notice
    name = re.search(r"^(_ZN12_GLOBAL__N_115conv_pws_kernel\w+):", asm, re.M).group(1)
    m = re.search(re.escape(name) + r":(.*?)\.Lfunc_end", asm, re.S)
    body = m.group(1).split("\n")
    loops, owner = isa_audit.mfma_loops([l.strip() for l in body])
    st = next(i for i, l in enumerate(body) if owner[i] in loops and l.strip().startswith("global_store"))
    broken = asm[:m.start(1)] + "\n".join(body[:st] + body[st + 1:]) + asm[m.end(1):]
    assert any("stores" in b for b in isa_audit.audit(broken))


def test_round4_entry_points_refuse_bad_arguments_without_launching(hip):
    """The entry points added in round 4 check their arguments on the host before anything is enqueued: a NULL operand or an operand type a
    kernel does not exist for comes back as TF_ERR_ARG / TF_ERR_UNSUPPORTED (the reference-side binding maps these to exceptions), never
    as a launch.  Runs without a GPU."""
    import ctypes as C
    l = hip.lib()
    ERR_ARG, ERR_UNSUPPORTED = -1, -3
    one = C.c_int(0)
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    # conv1 straight from the image: 2-byte operand types only, every operand required, the statistic epilogue needs its outputs
    assert l.tf_stem_conv(0, p, 1, 8, 8, p, 192, p, 0, None, None, None, None, None) == ERR_UNSUPPORTED          # TF_F32
    assert l.tf_stem_conv(1, None, 1, 8, 8, p, 192, p, 0, None, None, None, None, None) == ERR_ARG
    assert l.tf_stem_conv(1, p, 1, 8, 8, p, 100, p, 0, None, None, None, None, None) == ERR_ARG                    # ldw < 160
    assert l.tf_stem_conv(1, p, 1, 8, 8, p, 192, p, 8, None, None, None, C.byref(one), None) == ERR_ARG            # TF_EPI_STATS without rows
    assert l.tf_stem_conv(1, p, 1, 8, 8, p, 192, p, 5, None, None, None, None, None) == ERR_ARG                    # folded BN without vectors
    assert l.tf_stem_conv(1, p, 1, 8, 8, p, 192, p, 2, None, None, None, None, None) == ERR_UNSUPPORTED            # an epilogue it does not have
    assert l.tf_stem_wgrad(0, p, 1, 8, 8, p, None, None, None, None, p, None) == ERR_UNSUPPORTED
    assert l.tf_stem_wgrad(1, p, 1, 8, 8, None, None, None, None, None, p, None) == ERR_ARG
    assert l.tf_stem_wgrad(1, p, 1, 8, 8, p, p, None, None, None, p, None) == ERR_ARG                               # apply without coefficients
    assert l.tf_maxpool_bwd_stats(1, p, p, p, p, p, 1, 8, 8, 64, p, None, C.byref(one), None) == ERR_ARG
    assert l.tf_maxpool_bwd_stats(1, p, p, p, p, p, 1, 8, 8, 60, p, p, C.byref(one), None) == ERR_ARG                # C % 8
    assert l.tf_conv2d_wgrad_group(None, 1, None) == ERR_ARG
    # the executor context and the communicator: NULL handles are refused, not dereferenced
    assert l.tf_comm_rank(None) == -1 and l.tf_comm_world(None) == 0          # "no communicator": rank -1 of a world of 0
    assert l.tf_allreduce_bucket(None, p, 4, None) == ERR_ARG
