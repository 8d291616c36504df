"""Collapse one tools/profile.sh output directory into a small JSON summary.  All figures are PER STEP of the workload:
a step may launch the dominant kernel more than once (kNN at cfg3: the main launch, the column pieces of the rows beyond
a full round and their repair launch share one instantiation), so durations and counters of a kernel name are summed
over its launches and divided by the steps the profiled command ran (steps + warmup + the PCIe-inclusive step)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "prograph_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def newest(pattern):
    """one CSV per rerun accumulates in a trace directory: take the most recent"""
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def engine_kernel(name):
    return "pg_nsq_kernel" in name or "pg_mm_kernel" in name

def steps_run(bench_json):
    """steps the profiled bench.py command executed: timed + warmup (+ one PCIe-inclusive step when it reports one)"""
    try:
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        return int(b["steps"]) + int(b["warmup"]) + (1 if b.get("pcie_inclusive") else 0)
    except Exception:
        return None


out_dir, wl = sys.argv[1], sys.argv[2]
res = {"workload": wl, "kernel": None, "counters": {}, "command": f"python bench.py --workload {wl} --no-cpu-baseline --no-extra",
       "kernel_src_sha": kernel_src_sha(), "per": "step (sums over a kernel's launches of one step)"}
S = steps_run(f"{out_dir}/bench_trace.json")
for f in newest(f"{out_dir}/trace/**/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    S = S or max(int(r["Calls"]) for r in rows if engine_kernel(r["Name"]))
    for r in rows:
        if engine_kernel(r["Name"]) and (res["kernel"] is None or float(r["Percentage"]) > res["kernel"]["pct"]):
            res["kernel"] = {"name": r["Name"], "calls": int(r["Calls"]), "launches_per_step": int(r["Calls"]) / S,
                             "avg_ns": float(r["AverageNs"]) * int(r["Calls"]) / S, "avg_ns_per_launch": float(r["AverageNs"]),
                             "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    res["steps_in_trace"] = S
    res["all_kernels"] = [{"name": r["Name"][:80], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows]
pmc_files = []
for d in sorted(glob.glob(f"{out_dir}/pmc_*")):
    pmc_files += [(d, f) for f in newest(f"{d}/**/*_counter_collection.csv")]
for d, f in pmc_files:
    Sp = steps_run(f"{out_dir}/bench_{os.path.basename(d)[4:]}.json") or 5
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if res["kernel"] and r["Kernel_Name"] == res["kernel"]["name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["vgpr"] = int(r["VGPR_Count"]); res["sgpr"] = int(r["SGPR_Count"]); res["lds"] = int(r["LDS_Block_Size"])
            res["grid"] = max(res.get("grid", 0), int(r["Grid_Size"])); res["wg"] = int(r["Workgroup_Size"])
    for k, v in agg.items():
        res["counters"][k] = sum(v) / Sp
    # every kernel of a step: per-kernel counter sums per step
    allk = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        allk[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in allk.items():
        for cn, v in cs.items():
            res.setdefault("per_kernel_counters", {}).setdefault(kn[:80], {})[cn] = {"per_step": sum(v) / Sp, "launches_per_step": len(v) / Sp}
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
    # wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores (ours are narrower
    # scattered stores: uncalibrated, taken as is)
    res["hbm_bytes_per_step"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    res["hbm_note"] = "L2-miss fabric traffic incl. Infinity-Cache hits (not DRAM bytes); (2*FETCH_SIZE + WRITE_SIZE) KiB"
if "GRBM_GUI_ACTIVE" in c and res["kernel"]:
    res["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / res["kernel"]["avg_ns"]
if "SQ_INSTS_VALU" in c and res["kernel"]:
    res["valu_wave_instr_per_s"] = c["SQ_INSTS_VALU"] / (res["kernel"]["avg_ns"] * 1e-9)
    res["valu_issue_frac_of_peak"] = res["valu_wave_instr_per_s"] / (256 * 4 * 2.4e9 / 2)   # SIMD-32: 2 cycles per wave64 instruction
    res["valu_issue_frac_of_4cycle_rate"] = res["valu_wave_instr_per_s"] / (256 * 4 * 2.4e9 / 4)
if "SQ_INSTS_MFMA" in c and res["kernel"]:
    res["mfma_pipe_busy_frac"] = c["SQ_INSTS_MFMA"] * 32.0 / (256 * 4 * 2.4e9 * res["kernel"]["avg_ns"] * 1e-9)
if "per_kernel_counters" in res:
    res["valu_wave_instr_per_step_all_kernels"] = sum(v["SQ_INSTS_VALU"]["per_step"] for v in res["per_kernel_counters"].values() if "SQ_INSTS_VALU" in v)
if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE" in c:
    # busy cycles summed over the 256 texture-address units (one per CU) / (256 x kernel cycles): the vector-memory path
    res["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (256.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
    # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs x kernel cycles available
    res["valu_busy_frac_of_simd_cycles"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
if "TCC_HIT_sum" in c:
    res["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print(json.dumps(res, indent=1))
