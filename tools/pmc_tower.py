"""Mean per-launch counters of the tower kernel from a rocprofv3 --pmc counter_collection.csv, plus the figures derived
from them: effective clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs), share of all SIMD cycles the matrix pipes are busy
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles; 256 CUs x 4 SIMDs), split of the waves' cycles.
  python tools/pmc_tower.py <counter_collection.csv> [min_grid] [kernel name part, default k_tower]"""
import collections
import csv
import json
import sys

name = sys.argv[3] if len(sys.argv) > 3 else "k_tower"
rows = [r for r in csv.DictReader(open(sys.argv[1])) if name in r["Kernel_Name"]]
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 200
by = collections.defaultdict(list)
dur = {}
for r in rows:
    if int(r["Grid_Size"]) // int(r["Workgroup_Size"]) < min_grid:
        continue
    by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r.get("End_Timestamp"):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
mean = {k: sum(v) / len(v) for k, v in by.items()}
out = {"kernel": rows[0]["Kernel_Name"].split("(")[0], "launches": len(next(iter(by.values()))), "vgpr": rows[0].get("VGPR_Count"),
       "counters_per_launch": {k: round(v) for k, v in sorted(mean.items())}}
us = sum(dur.values()) / len(dur) if dur else None
out["launch_us_under_pmc"] = us and round(us, 1)
d = {}
if us and "GRBM_GUI_ACTIVE" in mean:
    d["effective_clock_GHz"] = round(mean["GRBM_GUI_ACTIVE"] / 8 / us / 1e3, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
        d["mfma_busy_fraction_of_all_simd_cycles"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
if "SQ_WAVE_CYCLES" in mean:
    w = mean["SQ_WAVE_CYCLES"]
    d["wave_cycles_split"] = {k: round(mean[k] / w, 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in mean}
out["derived"] = d
print(json.dumps(out, indent=1))
