"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter."""
import csv, sys, glob, os, collections
out = sys.argv[1]
for path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(row["Kernel_Name"][:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("==", os.path.relpath(path, out))
    for (kn, cn), v in sorted(acc.items()):
        if "scan_kernel" not in kn and "select" not in kn and "threshold" not in kn:
            continue
        print(f"{kn:48s} {cn:22s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
