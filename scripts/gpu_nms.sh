#!/bin/bash
mkdir -p gpurun_out/nms
python scripts/nms_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/nms/nms_bench.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/nm
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nm -o n -- python $GRAFT_REPO_ROOT/scripts/nms_bench.py > /dev/null 2>&1
t=$(find /tmp/nm -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/nms/nms_kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "nms_" in n:
        key = (n.split("nms_")[1].split("(")[0][:28], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort(); print(k, "n", len(v), "median us", round(v[len(v)//2], 1))
PY
