"""Static check of the fragment-ring RULE of pg_mm.h on the generated ISA: between an inline-assembly fragment load
(global_load_dwordx4 with a scalar base inside an ASMSTART block) and the s_waitcnt that covers it, no instruction may
read or write a register the load is still filling (the compiler does not know the load is in flight: a copy or spill
there would move stale data).  Every path through a kernel is walked (branches followed both ways), so the layout of the basic
blocks in the file does not matter.  usage: tools/check_ring_asm.py [G ...]   (compiles pg_nsq_inst.hip -S per group count)"""
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

def check_kernel(name, instrs, labels, g):
    """Walk every path through the kernel (worklist over (instruction, loads in flight in ISSUE order: a tuple of
    (line, registers), oldest first - across a loop's back edge line numbers do not tell the order)); returns (violations, loads)."""
    bad, seen, reported = 0, set(), set()
    work = [(0, ())]
    while work:
        pc, pend = work.pop()
        while pc < len(instrs):
            if (pc, pend) in seen:
                break
            seen.add((pc, pend))
            ln, t, in_asm = instrs[pc]
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
            if in_asm and t.startswith("global_load_dwordx4"):
                pend = pend + ((ln, tuple(sorted(regs(toks[0])))),)
                pc += 1
                continue
            if in_asm and t.startswith("s_waitcnt vmcnt("):
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                pend = pend[len(pend) - n:] if n and len(pend) > n else (pend if n else ())   # all but the youngest n loads are done
                pc += 1
                continue
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                pend = ()
                pc += 1
                continue
            touched = set().union(*[regs(x) for x in toks]) if toks else set()
            for src, rs in pend:
                hit = touched & set(rs)
                if hit and (ln, src) not in reported:
                    reported.add((ln, src))
                    bad += 1
                    print(f"G={g} {name} line {ln}: `{t}` touches v{sorted(hit)} while the load of line {src} is in flight")
            if t.startswith("s_endpgm"):
                break
            m = re.match(r"(s_branch|s_cbranch_\w+)\s+(\S+)", t)
            if m:
                tgt = labels.get(m.group(2))
                if tgt is not None:
                    work.append((tgt, pend))
                if m.group(1) == "s_branch":
                    break
            pc += 1
    nloads = sum(1 for ln, t, a in instrs if a and t.startswith("global_load_dwordx4"))
    return bad, nloads


bad = 0
for g in gs:
    out = os.path.join(tempfile.gettempdir(), f"ring_g{g}.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [f"-DPG_G={g}", "-S", "--cuda-device-only", "pg_nsq_inst.hip", "-o", out],
                          cwd=SRC, stderr=subprocess.DEVNULL)
    kernel, instrs, labels, in_asm, nloads, nk = None, [], {}, False, 0, 0
    for ln, line in enumerate(open(out), 1):
        t = line.strip()
        if t.startswith("_Z12pg_mm_kernel") and ":" in t and not t.endswith('"'):
            kernel, instrs, labels, in_asm = t.split(":")[0], [], {}, False
            nk += 1
            continue
        if not kernel:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r"(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(instrs); continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        instrs.append((ln, t, in_asm))
        if t.startswith("s_endpgm"):
            b, n = check_kernel(kernel, instrs, labels, g)
            bad += b; nloads += n
            kernel = None
    print(f"G={g}: {nk} pg_mm_kernel instances, {nloads} inline fragment loads checked")
print("RULE violations:", bad)
sys.exit(1 if bad else 0)
