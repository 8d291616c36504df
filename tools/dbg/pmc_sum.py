"""Mean counter values per launch of the kernels whose name contains argv[2] in a rocprofv3 counter_collection CSV (argv[1])."""
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"{k:32s} {sum(v) / len(v):14.4e}  (n={len(v)})")
