"""Per-kernel-name totals per training step from a rocprofv3 kernel trace (steps delimited by targets_kernel<0>)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:44]
rows.sort(key=lambda r: r["s"])
st = [i for i, r in enumerate(rows) if r["n"].startswith("targets_kernel<0>")]
a, b = st[-2], st[-1]
step = rows[a:b]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in step:
    agg[r["n"]][0] += (r["e"] - r["s"]) / 1e3; agg[r["n"]][1] += 1
print(f"step wall {(rows[b]['s'] - rows[a]['s']) / 1e3:.0f} us, kernels {len(step)}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"  {k:46s} {v[0]:9.1f} us  n={v[1]:4d}  avg {v[0] / v[1]:7.1f}")
