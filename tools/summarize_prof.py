"""Collapse one tools/profile.sh output directory into a small JSON summary (per-launch averages)."""
import collections
import csv
import glob
import json
import sys

out_dir, wl = sys.argv[1], sys.argv[2]
res = {"workload": wl, "kernel": None, "counters": {}, "command": f"python bench.py --workload {wl} --no-cpu-baseline"}
for f in glob.glob(f"{out_dir}/trace/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "pg_nsq_kernel" in r["Name"]:
            res["kernel"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                             "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    res["all_kernels"] = [{"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows]
for f in glob.glob(f"{out_dir}/pmc_*/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pg_nsq_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["vgpr"] = int(r["VGPR_Count"]); res["sgpr"] = int(r["SGPR_Count"]); res["lds"] = int(r["LDS_Block_Size"])
            res["grid"] = int(r["Grid_Size"]); res["wg"] = int(r["Workgroup_Size"])
    for k, v in agg.items():
        res["counters"][k] = sum(v) / len(v)
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
    # wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores (ours are narrower
    # scattered stores: uncalibrated, taken as is)
    res["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    res["hbm_note"] = "L2-miss fabric traffic incl. Infinity-Cache hits (not DRAM bytes); (2*FETCH_SIZE + WRITE_SIZE) KiB"
if "GRBM_GUI_ACTIVE" in c and res["kernel"]:
    res["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / res["kernel"]["avg_ns"]
if "SQ_INSTS_VALU" in c and res["kernel"]:
    res["valu_wave_instr_per_s"] = c["SQ_INSTS_VALU"] / (res["kernel"]["avg_ns"] * 1e-9)
    res["valu_issue_frac_of_peak"] = res["valu_wave_instr_per_s"] / (256 * 4 * 2.4e9 / 4)
if "TCC_HIT_sum" in c:
    res["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print(json.dumps(res, indent=1))
