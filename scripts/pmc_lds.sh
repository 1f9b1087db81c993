#!/bin/bash
# one SQ pass focused on the LDS: bash scripts/pmc_lds.sh <script.py> [kernel regex]   -> gpurun_out/pmc/lds_summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p "$R/gpurun_out/pmc"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_lds && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --output-format csv -d /tmp/pmc_lds -o p -- python $R/$1 > /tmp/pmc_lds.log 2>&1 ); echo "pmc exit $?"
f=$(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1)
python - "$f" "${2:-wgrad}" <<'PY' | tee "$R/gpurun_out/pmc/lds_summary.txt"
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    if not re.search(sys.argv[2], n): continue
    g = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
    agg[(n[:48], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[(n[:48], g)]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(n, "grid", g, " ".join(f"{k}={v:.4g}" for k, v in sorted(m.items())))
PY
