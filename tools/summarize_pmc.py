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
KEEP = ("stream_scan_kernel", "stream_scan_long_kernel", "stream_sample_kernel", "stream_rank_kernel", "scan_fused_kernel", "scan_kernel", "scan_mx_kernel", "scan_mq_kernel", "scan_mq8_kernel", "scan_lq_kernel", "boot_mq_kernel", "embed_scan_kernel", "embed_px_kernel", "embed_mx_kernel", "rank_sort", "select", "threshold")
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
    cands = [kn for (kn, cn) in means if cn == "FETCH_SIZE" and ("stream_scan_kernel" in kn or "scan_fused_kernel" in kn or "scan_mx_kernel" in kn
                                                                  or "scan_kernel<20, true, 1>" in kn or "scan_kernel<20,true,1>" in kn)]
    if cands:
        # (bench.py's default run holds the overlap steps AND, for comparison, the same steps as fused launches: the
        #  headline's dominant kernel is the overlap scan)
        pref = [k for k in cands if "stream_scan_kernel" in k]
        kn = pref[0] if pref else max(cands, key=lambda k: means[(k, "FETCH_SIZE")])
        fetch = means[(kn, "FETCH_SIZE")]
        write = means.get((kn, "WRITE_SIZE"), 0.0)
        R, T, W, h, k, B = 32768, 4096, 20, 20, 1024, 1
        extra = {}
        for other in ("stream_sample_kernel", "stream_rank_kernel"):     # the small launches of an overlap step, for the record
            for (k2, c2) in means:
                if other in k2 and c2 == "FETCH_SIZE":
                    extra[other + "_hbm_bytes_per_launch"] = int(2 * means[(k2, "FETCH_SIZE")] * 1024 + means.get((k2, "WRITE_SIZE"), 0.0) * 1024)
        j = {"workload": f"R={R},T={T},W={W},h={h},k={k},B={B}", "kernel": kn, "round": 6, **extra,
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

if "--mq" in sys.argv:
    # the batched scan (bench.py --queries 512): busy fractions of the matrix cores and the vector ALUs of scan_mq_kernel, and
    # its HBM traffic, for bench.py's "mfma+valu" roofline object
    dst = sys.argv[sys.argv.index("--mq") + 1]
    kn = next((k for (k, c) in means if ("scan_mq8_kernel" in k or "scan_mq_kernel" in k) and c == "GRBM_GUI_ACTIVE"), None)
    if kn:
        cyc = means[(kn, "GRBM_GUI_ACTIVE")] / 8.0                 # summed over the 8 XCDs -> shader cycles of the launch
        nsimd = 1024.0
        mfma = means.get((kn, "SQ_VALU_MFMA_BUSY_CYCLES"))
        valu = means.get((kn, "SQ_ACTIVE_INST_VALU"))              # quad-cycles (MI355X_MICROARCH.md)
        fetch, write = means.get((kn, "FETCH_SIZE")), means.get((kn, "WRITE_SIZE"), 0.0)
        R, T, W, h, k, B = 32768, 4096, 20, 20, 1024, 512
        j = {"workload": f"R={R},T={T},W={W},h={h},k={k},B={B}", "kernel": kn, "round": 6, "launch_cycles": round(cyc),
             "matrix_core_busy_frac": round(mfma / (nsimd * cyc), 4) if mfma else None,
             "valu_busy_frac": round(4.0 * valu / (nsimd * cyc), 4) if valu else None,
             "SQ_INSTS_MFMA": means.get((kn, "SQ_INSTS_MFMA")), "SQ_INSTS_VALU": means.get((kn, "SQ_INSTS_VALU")),
             "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024) if fetch else None,
             "units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs; SQ_ACTIVE_INST_VALU: quad-cycles summed over the SIMDs "
                      "(it includes the issue of the MFMAs themselves); launch cycles = GRBM_GUI_ACTIVE / 8 XCDs; FETCH_SIZE x 2 "
                      "(gfx950 correction) + WRITE_SIZE",
             "source": "rocprofv3 --pmc passes (separate runs, no trace options) -- python bench.py --steps 2 --warmup 1 --queries 512"}
        with open(dst, "w") as f:
            json.dump(j, f, indent=1)
        print("wrote", dst, j)
