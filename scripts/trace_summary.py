"""Summarise a rocprofv3 kernel_trace.csv: average duration per (kernel, grid, lds) -- true GPU durations, no host overhead."""
import csv, os, re, sys, collections
NL = int(os.environ.get("NAMELEN", "56"))
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    if pat and not re.search(pat, n):
        continue
    k = (n[:NL], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v2 = sorted(v)[: max(1, len(v) * 3 // 4)]          # drop the slowest quarter (warm-up / clock ramp)
    print(f"{k[0]:{NL + 2}s} grid={k[1]:6d} n={len(v):4d} avg={sum(v) / len(v):8.1f}us trimmed={sum(v2) / len(v2):8.1f}us min={min(v):8.1f}us")
