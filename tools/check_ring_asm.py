"""Static check of the fragment-ring RULE of pg_mm.h on the generated ISA: between an inline-assembly fragment load
(global_load_dwordx4 with a scalar base inside an ASMSTART block) and the s_waitcnt that covers it, no instruction may
read or write a register the load is still filling (the compiler does not know the load is in flight: a copy or spill
there would move stale data).  usage: tools/check_ring_asm.py [G ...]   (compiles pg_nsq_inst.hip -S per group count)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "prograph_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-pass-failed -Wno-unused-variable -mllvm -amdgpu-mfma-vgpr-form=1".split()
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
gs = [int(a) for a in sys.argv[1:] if a.isdigit()] or list(range(1, 9))

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

bad = 0
for g in gs:
    out = os.path.join(tempfile.gettempdir(), f"ring_g{g}.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [f"-DPG_G={g}", "-S", "--cuda-device-only", "pg_nsq_inst.hip", "-o", out],
                          cwd=SRC, stderr=subprocess.DEVNULL)
    kernel, pending, in_asm, nloads, nk = None, {}, False, 0, 0
    for ln, line in enumerate(open(out), 1):
        t = line.strip()
        if t.startswith("_Z12pg_mm_kernel") and t.endswith(":") is False and ":" in t:
            kernel, pending = t.split(":")[0], {}
            nk += 1
        if not kernel:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("s_endpgm"):
            kernel = None; continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if in_asm and t.startswith("global_load_dwordx4"):
            for r in regs(toks[0]):
                pending[r] = ln
            nloads += 1
            continue
        if in_asm and t.startswith("s_waitcnt vmcnt("):
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            # loads land in order: all but the youngest n loads (4 registers each) are done
            keep = sorted(set(pending.values()))[-n:] if n else []
            pending = {r: l for r, l in pending.items() if l in keep}
            continue
        if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            pending = {}
            continue
        touched = set().union(*[regs(x) for x in toks]) if toks else set()
        hit = touched & set(pending)
        if hit:
            bad += 1
            print(f"G={g} {kernel} line {ln}: `{t}` touches v{sorted(hit)} while the load of line {pending[min(hit)]} is in flight")
    print(f"G={g}: {nk} pg_mm_kernel instances, {nloads} inline fragment loads checked")
print("RULE violations:", bad)
sys.exit(1 if bad else 0)
