"""Last launches of a rocprofv3 kernel trace CSV: name, start, duration, gap to the previous launch (microseconds)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
t0 = int(rows[-n]["Start_Timestamp"]); prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(r["Kernel_Name"][:64].ljust(64), "start %9.1f dur %8.1f gap %6.1f grid %s" % (s / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev is not None else 0), r.get("Grid_Size")))
    prev = e
