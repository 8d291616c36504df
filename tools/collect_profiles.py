"""Copy the judged artefacts of tools/profile.sh runs (gpurun_out/prof/<tag>_<wl>/) into profiles/ and
rebuild profiles/pmc_summary.json (read by bench.py for roofline.traffic / valu_frac).
usage: tools/collect_profiles.py r02 cfg3 cfg2 cfg5 cfg3d"""
import glob, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wls = sys.argv[1], sys.argv[2:]
pmc_path = os.path.join(root, "profiles", "pmc_summary.json")
pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
for wl in wls:
    src = os.path.join(root, "gpurun_out", "prof", f"{tag}_{wl}")
    summ = json.load(open(os.path.join(src, "summary.json")))
    shutil.copy(os.path.join(src, "summary.json"), os.path.join(root, "profiles", f"{tag}_{wl}_rocprof_summary.json"))
    shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(root, "profiles", f"{tag}_{wl}_bench_under_rocprof.json"))
    ks = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:    # a trace directory accumulates one CSV per rerun: the newest belongs to this summary
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(root, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
    b = os.path.join(root, "gpurun_out", f"bench_{wl}.json")
    if os.path.exists(b):
        shutil.copy(b, os.path.join(root, "profiles", f"{tag}_bench_{wl}.json"))
    pmc[wl] = {
        "hbm_bytes_per_step": summ.get("hbm_bytes_per_step"),
        "kernel_avg_ns": summ["kernel"]["avg_ns"],
        "kernel_name": summ["kernel"]["name"][:60],
        "kernel_src_sha": summ.get("kernel_src_sha"),
        "valu_issue_frac": summ.get("valu_issue_frac_of_peak"),
        "valu_wave_instr_per_step": summ["counters"].get("SQ_INSTS_VALU"),
        "valu_wave_instr_per_step_all_kernels": summ.get("valu_wave_instr_per_step_all_kernels"),
        "salu_instr_per_step": summ["counters"].get("SQ_INSTS_SALU"),
        "mfma_instr_per_step": summ["counters"].get("SQ_INSTS_MFMA"),
        "wait_any_frac_of_wave_cycles": (summ["counters"].get("SQ_WAIT_ANY", 0) / summ["counters"]["SQ_WAVE_CYCLES"]) if summ["counters"].get("SQ_WAVE_CYCLES") else None,
        "clock_ghz": summ.get("clock_ghz"),
        "l2_hit_rate": summ.get("l2_hit_rate"),
        "ta_busy_frac": summ.get("ta_busy_frac"),
        "valu_busy_frac_of_simd_cycles": summ.get("valu_busy_frac_of_simd_cycles"),
        "mfma_pipe_busy_frac": summ.get("mfma_pipe_busy_frac"),
        "source": f"profiles/{tag}_{wl}_rocprof_summary.json",
        "note": summ.get("hbm_note"),
    }
json.dump(pmc, open(pmc_path, "w"), indent=1)
print(json.dumps(pmc, indent=1))
