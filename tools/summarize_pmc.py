"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel mean of each counter.
With --traffic OUT.json also writes the HBM traffic of the dominant scan kernel per launch
(FETCH_SIZE / WRITE_SIZE passes), corrected as MI355X_MICROARCH.md's HBM section prescribes."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KEEP = ("scan_fused_kernel", "scan_kernel", "scan_mx_kernel", "scan_mq_kernel", "boot_mq_kernel", "embed_scan_kernel", "embed_px_kernel", "embed_mx_kernel", "rank_sort", "select", "threshold")
means = {}
for path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[(row["Kernel_Name"][:56], row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("==", os.path.relpath(path, out))
    for (kn, cn), v in sorted(acc.items()):
        if not any(s in kn for s in KEEP):
            continue
        means[(kn, cn)] = sum(v) / len(v)
        print(f"{kn:56s} {cn:26s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")

if "--traffic" in sys.argv:
    dst = sys.argv[sys.argv.index("--traffic") + 1]
    # the dominant kernel: the full scan (FILTER instantiation <..,1> of scan_kernel, or scan_mx_kernel)
    cands = [kn for (kn, cn) in means if cn == "FETCH_SIZE" and ("scan_fused_kernel" in kn or "scan_mx_kernel" in kn or "scan_kernel<20, true, 1>" in kn
                                                                  or "scan_kernel<20,true,1>" in kn)]
    if cands:
        kn = max(cands, key=lambda k: means[(k, "FETCH_SIZE")])
        fetch = means[(kn, "FETCH_SIZE")]
        write = means.get((kn, "WRITE_SIZE"), 0.0)
        R, T, W, h, k, B = 32768, 4096, 20, 20, 1024, 1
        j = {"workload": f"R={R},T={T},W={W},h={h},k={k},B={B}", "kernel": kn, "round": 2,
             "FETCH_SIZE_KiB_per_launch": round(fetch, 1), "WRITE_SIZE_KiB_per_launch": round(write, 1),
             "correction": "gfx950 rocprofv3 FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams: read bytes = "
                           "2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
             "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
             "algorithmic_bytes_per_launch": R * T * 4 + B * W * 4 + B * k * 12,
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace options) -- python bench.py "
                       "--steps 5 --warmup 2; mean over the launches"}
        with open(dst, "w") as f:
            json.dump(j, f, indent=1)
        print("wrote", dst, j["hbm_bytes_per_launch"], "bytes per launch vs algorithmic", j["algorithmic_bytes_per_launch"])
