#!/bin/bash
mkdir -p gpurun_out/r3e
python scripts/floor.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e/floor.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fl
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fl -o f -- python $GRAFT_REPO_ROOT/scripts/floor.py > /dev/null 2>&1
f=$(find /tmp/fl -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3e/floor_kernel_stats.csv; head -5 "$f"
t=$(find /tmp/fl -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "probe_kernel" in r["Kernel_Name"]:
        g = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
        agg[(g, r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), r["Kernel_Name"][-30:])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    v.sort(); print(k, "n", len(v), "median ns", v[len(v)//2], "min", v[0])
PY
