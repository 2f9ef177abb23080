"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter."""
import csv, sys, glob, os, collections
out = sys.argv[1]
for path in sorted(glob.glob(os.path.join(out, "prof_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("==", path)
    for (kn, cn), v in sorted(acc.items()):
        print(f"{kn:60s} {cn:12s} n={len(v):4d} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}")
