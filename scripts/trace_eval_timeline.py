"""Per-queue view of ONE evaluation pyramid from a rocprofv3 kernel trace of `bench.py --eval-only`: the image is delimited by two consecutive
nms_rank launches of the single-image leg; per queue busy time, idle gaps, first / last kernel; union busy of the GPU."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:44]
    r["q"] = r.get("Queue_Id", "0")
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if r["n"].startswith("nms_rank")]
# windows between consecutive NMS launches that look like one image (shortest typical span)
spans = [(rows[b]["s"] - rows[a]["s"], a, b) for a, b in zip(marks, marks[1:])]
spans = [s for s in spans if 3e6 < s[0] < 12e6]
spans.sort()
span, a, b = spans[len(spans) // 4]
img = rows[a + 1:b + 1]
t0 = img[0]["s"]
print(f"image wall {span / 1e3:.0f} us, {len(img)} kernels")
byq = collections.defaultdict(list)
for r in img: byq[r["q"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs) / 1e3
    gaps = [(y["s"] - x["e"]) / 1e3 for x, y in zip(rs, rs[1:]) if y["s"] > x["e"]]
    print(f"queue {q}: {len(rs)} kernels, busy {busy:.0f} us, first start {(rs[0]['s'] - t0) / 1e3:.0f}, last end {(rs[-1]['e'] - t0) / 1e3:.0f}, idle gaps sum {sum(gaps):.0f} us "
          f"(n={len(gaps)}, median {sorted(gaps)[len(gaps) // 2] if gaps else 0:.1f}, >10us: {sum(1 for g in gaps if g > 10)} = {sum(g for g in gaps if g > 10):.0f} us)")
    big = sorted(((y["s"] - x["e"]) / 1e3, x["n"], y["n"], (x["e"] - t0) / 1e3) for x, y in zip(rs, rs[1:]))[-4:]
    for g, xn, yn, at in reversed(big):
        print(f"      gap {g:6.1f} us at {at:7.0f}: {xn} -> {yn}")
ev = sorted([(r["s"], 1) for r in img] + [(r["e"], -1) for r in img])
depth, last, busy = 0, t0, 0
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f"GPU busy (union over queues) {busy / 1e3:.0f} us of {span / 1e3:.0f}")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in img: agg[r["n"]][0] += (r["e"] - r["s"]) / 1e3; agg[r["n"]][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:34]:
    print(f"    {k:46s} {v[0]:8.1f} us n={v[1]:4d} avg {v[0] / v[1]:6.1f}")
