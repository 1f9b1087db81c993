"""ISA audit of csrc/conv_pws.hip, run by build.py whenever the file is compiled and by tests/test_cabi.py.

Every wave of conv_pws walks its own 16-pixel tiles through a private two-slot LDS ring with NO block barrier in the loop: the orderings a barrier
would have given are kept by ONE hand-counted `s_waitcnt vmcnt(NST + NIX + NIA)` per tile, which is only right while hipcc emits exactly NST output
stores and NIX + NIA LDS-DMA instructions per tile and none of its own waits on the VM counter.  `audit(asm)` checks every instantiation:
  * no scratch; every global_load_lds sits in inline asm; exactly one MFMA loop;
  * every wait on the VM counter inside the loop is a hand-written one, and its count is (stores + DMAs of the loop body) -- or the DMA count alone
    for the first tile --, with stores + 2 x DMAs <= 63 (the 6-bit counter);
  * no DMA of the loop is issued with an LDS access of the wave outstanding.
Returns the list of violations (empty = pass)."""
import re
import subprocess

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17"]


def emit_isa(hipcc, src, out, extra=()):
    r = subprocess.run([hipcc] + HIPCC_FLAGS + list(extra) + ["-S", "--cuda-device-only", src, "-o", out], capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for " + src + ":\n" + r.stderr[-3000:])
    with open(out) as f:
        return f.read()


def mfma_loops(lines):
    """(headers of the loops that hold an MFMA, line -> header of the loop it belongs to) from hipcc's block comments."""
    owner, cur = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"\.L(BB\d+_\d+):\s*;(.*)", l)
        if m:
            h = re.search(r"Header=(BB\d+_\d+)", m.group(2))
            cur = h.group(1) if h else (m.group(1) if "Loop Header" in m.group(2) else None)
        elif re.match(r"\.L(BB\d+_\d+):", l):
            cur = None
        owner[i] = cur
    return {owner[i] for i, l in enumerate(lines) if l.startswith("v_mfma")} - {None}, owner      # (None: the first tile, peeled in front of the loop)


def audit(asm, min_kernels=30):
    bad = []
    names = re.findall(r"^(_ZN12_GLOBAL__N_115conv_pws_kernel\w+):", asm, re.M)
    if len(names) < min_kernels:
        bad.append(f"only {len(names)} conv_pws instantiations found")
    for name in names:
        body = re.search(re.escape(name) + r":(.*?)\.Lfunc_end", asm, re.S).group(1)
        if "scratch_" in body:
            bad.append(f"{name}: scratch")
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
        if not dmas or not all(i in hand for i in dmas):
            bad.append(f"{name}: a global_load_lds outside inline asm (or none at all)")
        loops, owner = mfma_loops(lines)
        if len(loops) != 1:
            bad.append(f"{name}: {len(loops)} MFMA loops")
            continue
        loop = [i for i in range(len(lines)) if owner[i] in loops]
        waits = [i for i in loop if lines[i].startswith("s_waitcnt") and "vmcnt" in lines[i]]
        if not waits or not all(i in hand for i in waits):
            bad.append(f"{name}: a compiler-made vmcnt wait in the tile loop: {[lines[i] for i in waits if i not in hand]}")
        n_dma = sum(1 for i in loop if lines[i].startswith("global_load_lds"))
        n_st = sum(1 for i in loop if lines[i].startswith("global_store") or lines[i].startswith("buffer_store"))
        counts = {int(re.search(r"vmcnt\((\d+)\)", lines[i]).group(1)) for i in waits if i in hand}
        if not counts <= {n_dma, n_st + n_dma} or (n_st + n_dma) not in counts:
            bad.append(f"{name}: tile loop holds {n_st} stores + {n_dma} DMAs but waits for vmcnt{sorted(counts)}")
        if n_st + 2 * n_dma > 63:
            bad.append(f"{name}: {n_st} + 2 x {n_dma} instructions in flight exceed the 6-bit VM counter")
        pend = 0
        for i in loop + loop:                      # two trips around the loop body: LDS accesses outstanding at every DMA issue
            l = lines[i]
            if re.match(r"ds_(read|write|load|store)", l):
                pend += 1
            elif l.startswith("s_waitcnt"):
                w = re.search(r"lgkmcnt\((\d+)\)", l)
                if w:
                    pend = min(pend, int(w.group(1)))
            elif l.startswith("global_load_lds") and pend:
                bad.append(f"{name}: DMA issued with {pend} LDS accesses outstanding (line {i})")
                break
    return bad
