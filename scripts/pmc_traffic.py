"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both reported in KiB per dispatch).
gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests at 64 bytes for
wide coalesced reads (16 B/lane global_load and global_load_lds alike) -> doubled.  WRITE_SIZE is taken as reported."""
import csv, json, os, re, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tiny-faces-pytorch_amd")]

def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            agg[n][0] += 1
            agg[n][1] += float(r["Counter_Value"])
    return agg

fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"unit": "bytes per launch (average over the run)", "fetch_correction": "FETCH_SIZE x 2 (gfx950 wide-read correction)", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    nf, kb = fetch[k]
    nw, wkb = write.get(k, [0, 0.0])
    out["kernels"][k] = {"launches": nf, "fetch_bytes": round(kb * 2 * 1024 / max(nf, 1)), "write_bytes": round(wkb * 1024 / max(nw, 1)),
                         "hbm_bytes": round(kb * 2 * 1024 / max(nf, 1) + wkb * 1024 / max(nw, 1))}
try:       # r6: stamp the file with the library it was measured on (bench.py prints traffic_stale when the loaded one differs)
    from tinyfaces import _hip
    out["library"] = _hip.identity()
except Exception as e:
    out["library"] = {"error": repr(e)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{k[:60]:60s} n={v['launches']:5d} fetch={v['fetch_bytes'] / 1e6:8.2f} MB write={v['write_bytes'] / 1e6:8.2f} MB")
