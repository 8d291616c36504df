"""Start / duration of every row pass of one kNN launch (debug build with -DPG_MM_STATS): the shape of the grid's tail.
usage: pass_timeline.py [cfg3|dense]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PROGRAPH_HIP_LIB"] = os.path.join(ROOT, "prograph_amd", "libprograph_hip_stats.so")
os.environ["PG_ENGINE"] = "mfma"
import numpy as np, torch
from prograph_amd import _native as nat, synth
lib = nat.lib()
for name in (sys.argv[1:] or ["dense"]):
    tok = synth.clustered_tokens(200000, 64, members=200000) if name == "dense" else synth.clustered_tokens(200000, 64)
    p = nat.pack(torch.from_numpy(tok), bits=5)
    out = (torch.empty((200000, 16), dtype=torch.int32, device=p.buf.device), torch.empty((200000, 16), dtype=torch.uint8, device=p.buf.device))
    nat.knn_graph(p, p, 16, out=out); torch.cuda.synchronize()
    lib.pg_debug_stats(None, 1)
    nat.knn_graph(p, p, 16, out=out); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (24 + 2 * 65536))()
    lib.pg_debug_stats(buf, 3)
    raw = np.frombuffer(buf, dtype=np.uint64)[24:].reshape(-1, 2)
    raw = raw[raw[:, 0] != 0]
    cand = ((raw[:, 1] >> np.uint64(24)) & np.uint64(0xFFFFF)).astype(np.float64); exact = ((raw[:, 1] >> np.uint64(44)) & np.uint64(0x3FF)).astype(np.float64); flushes = (raw[:, 1] >> np.uint64(54)).astype(np.float64)
    a = np.stack([raw[:, 0].astype(np.float64), (raw[:, 1] & np.uint64(0xFFFFFF)).astype(np.float64)], axis=1)
    t0 = a[:, 0].min()
    start, dur = (a[:, 0] - t0), a[:, 1]
    end = start + dur
    tot = end.max()
    start, dur, end, tot = start / 100.0, dur / 100.0, end / 100.0, tot / 100.0     # 100 MHz -> microseconds
    print(f"== {name}: kernel span {tot:.1f} us; pass duration min/median/mean/max = {dur.min():.1f} {np.median(dur):.1f} {dur.mean():.1f} {dur.max():.1f}")
    print("   starts: first round (start < 1% of span):", int((start < 0.01 * tot).sum()), " later:", int((start >= 0.01 * tot).sum()))
    for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99): print(f"   duration q{q}: {np.quantile(dur, q):.1f}   end q{q}: {np.quantile(end, q) / tot:.3f} of span")
    late = start >= 0.01 * tot
    print(f"   first-round durations mean {dur[~late].mean():.1f}, later passes mean {dur[late].mean():.1f}; later starts from {start[late].min() / tot:.3f} to {start[late].max() / tot:.3f} of span")
    o = np.argsort(dur)
    for lab, idx in (("fastest 5%", o[:len(o) // 20]), ("middle", o[len(o) * 9 // 20:len(o) * 11 // 20]), ("slowest 5%", o[-(len(o) // 20):]), ("slowest 20", o[-20:])):
        print(f"   {lab:12s}: duration {dur[idx].mean():8.1f} us  folded candidates {cand[idx].mean():9.0f}  exact half-tiles {exact[idx].mean():6.1f}  flushes {flushes[idx].mean():6.1f}  started at {start[idx].mean() / tot:.2f}")
    occ = [( (start <= x * tot) & (end > x * tot)).sum() for x in np.linspace(0.02, 0.98, 25)]
    print("   waves in flight along the span:", " ".join(str(int(o)) for o in occ))
