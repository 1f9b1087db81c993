"""Critical-path view of ONE training step from a rocprofv3 kernel trace (`scripts/gpu_job.sh timeline`): per queue busy time, idle gaps,
overlap between the two streams, and the phases of the step (forward / backward chain / tail)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:40]
    r["q"] = r.get("Queue_Id", "0")
rows.sort(key=lambda r: r["s"])
st = [i for i, r in enumerate(rows) if r["n"].startswith("targets_kernel<0>")]
a, b = st[-2], st[-1]
step = rows[a:b]
t0 = step[0]["s"]
wall = (rows[b]["s"] - t0) / 1e3
print(f"step wall {wall:.0f} us, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step:
    byq[r["q"]].append(r)
mainq = step[0]["q"]
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs) / 1e3
    gaps = [(y["s"] - x["e"]) / 1e3 for x, y in zip(rs, rs[1:])]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}{' (main)' if q == mainq else ''}: {len(rs)} kernels, busy {busy:.0f} us, first start {(rs[0]['s']-t0)/1e3:.0f}, last end {(rs[-1]['e']-t0)/1e3:.0f}, "
          f"idle gaps: n={len(pos)} sum={sum(pos):.0f} us, median={sorted(pos)[len(pos)//2] if pos else 0:.1f}, >20us: {sum(1 for g in pos if g > 20)} ({sum(g for g in pos if g > 20):.0f} us)")
main = byq[mainq]
# phases on the main queue: forward = up to the first crit kernel; backward = from upsample_add_bwd to the last kernel before sgd
def first(pred):
    for i, r in enumerate(main):
        if pred(r["n"]):
            return i
    return None
ic = first(lambda n: n.startswith("crit_ohem"))
ib = first(lambda n: n.startswith("upsample_add_bwd"))
isg = first(lambda n: n.startswith("sgd_kernel"))
if None not in (ic, ib, isg):
    print(f"forward  : {(main[ic]['s'] - t0)/1e3:.0f} us ({ic} kernels)")
    print(f"criterion: {(main[ib]['s'] - main[ic]['s'])/1e3:.0f} us")
    print(f"backward : {(main[isg]['s'] - main[ib]['s'])/1e3:.0f} us ({isg - ib} kernels on the main queue)")
    print(f"tail     : {(rows[b]['s'] - main[isg]['s'])/1e3:.0f} us")
    bw = main[ib:isg]
    busy = sum(r["e"] - r["s"] for r in bw) / 1e3
    print(f"backward main-queue busy {busy:.0f} us, idle {(main[isg]['s'] - main[ib]['s'])/1e3 - busy:.0f} us; last 5 main kernels before sgd:")
    for r in bw[-5:]:
        print(f"    {r['n']:40s} start {(r['s']-t0)/1e3:8.0f} dur {(r['e']-r['s'])/1e3:6.1f}")
    for q, rs in byq.items():
        if q != mainq:
            print(f"  side queue {q}: last end {(rs[-1]['e']-t0)/1e3:.0f} us vs sgd start {(main[isg]['s']-t0)/1e3:.0f} us")
# biggest idle gaps on the main queue with the kernels around them
gl = sorted(((y["s"] - x["e"]) / 1e3, x["n"], y["n"], (x["e"] - t0) / 1e3) for x, y in zip(main, main[1:]))[-12:]
print("largest main-queue gaps:")
for g, xn, yn, at in reversed(gl):
    print(f"    {g:7.1f} us at {at:8.0f}: {xn} -> {yn}")
agg = collections.defaultdict(lambda: [0.0, 0])
for r in main:
    agg[r["n"]][0] += (r["e"] - r["s"]) / 1e3; agg[r["n"]][1] += 1
print("main-queue kernel totals:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"    {k:42s} {v[0]:8.1f} us n={v[1]:4d} avg {v[0]/v[1]:6.1f}")
# r4: who are the small blit dispatches?  (queue, kernel before, kernel after) of every copyBuffer / fillBuffer of the step, aggregated
blit = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if r["n"].startswith("__amd_rocclr"):
            prev = rs[i - 1]["n"][:34] if i else "-"
            nxt = rs[i + 1]["n"][:34] if i + 1 < len(rs) else "-"
            blit[(q == mainq, r["n"][13:], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), prev, nxt)] += 1
print("blit dispatches of the step (main queue?, kind, grid, previous kernel, next kernel): count")
for k, v in sorted(blit.items(), key=lambda kv: -kv[1])[:40]:
    print(f"    {str(k[0]):5s} {k[1]:18s} grid={k[2]:6d} {k[3]:34s} -> {k[4]:34s} x{v}")
# r5: the whole step, kernel by kernel (queue, start, duration, grid, workgroup, name): the single-launch outliers live here
print("sequence (queue | start us | dur us | grid x wg | kernel):")
for r in step:
    full = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:90]
    print(f"  {'M' if r['q'] == mainq else 's'} {(r['s']-t0)/1e3:8.1f} {(r['e']-r['s'])/1e3:7.1f} {int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0):8d}x{int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0)) or 0):4d} {full}")
