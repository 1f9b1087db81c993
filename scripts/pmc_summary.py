"""Per (kernel, grid) averages of one rocprofv3 --pmc counter_collection.csv (SQ pass) with the ratios that matter for an MFMA kernel."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and not re.search(sys.argv[2], n):
        continue
    g = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) if "Grid_Size" in r else 0
    agg[(n[:52], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[(n[:52], g)]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): upper-clock estimate of the matrix-pipe occupancy (profiled passes clock lower)")
print("kernel | grid | dur_us(profiled) | mfma_util | wave_cycles(quad) | active% | wait_any% | wait_inst% | mfma_busy cycles | mfma_busy/(4*wave_cycles) | lds_conflict/lds_active")
for (n, g), c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    du = m.get("_dur_us", 0) or 1
    print(f"{n:52s} | {g:6d} | {du:7.1f} | {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*du*2400):5.3f} | {wc:12.0f} | {100*m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f} | {100*m.get('SQ_WAIT_ANY',0)/wc:5.1f} | {100*m.get('SQ_WAIT_INST_ANY',0)/wc:5.1f} | "
          f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):12.0f} | {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc):6.3f} | {m.get('SQ_LDS_BANK_CONFLICT',0)/max(1,m.get('SQ_LDS_IDX_ACTIVE',1)):6.3f}")
