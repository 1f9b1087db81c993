"""Average duration per kernel name in two rocprofv3 kernel traces (A = reference, B = variant): which kernels stretch?"""
import csv, re, sys, collections
def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:44]
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for n in a:
    if n in b and len(a[n]) >= 3:
        ma, mb = sum(a[n]) / len(a[n]), sum(b[n]) / len(b[n])
        rows.append((sum(b[n]) - sum(a[n]) * len(b[n]) / len(a[n]), n, len(a[n]), ma, mb))
rows.sort(reverse=True)
print(f"{'kernel':46s} {'n':>5s} {'A us':>8s} {'B us':>8s} {'B/A':>5s}")
for _, n, k, ma, mb in rows[:28]:
    print(f"{n:46s} {k:5d} {ma:8.1f} {mb:8.1f} {mb / ma:5.2f}")
