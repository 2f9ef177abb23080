"""Summarise a rocprofv3 --kernel-trace csv of tools/two_stream_probe.py-like runs: per kernel name count / mean duration,
and for the scan kernels of the steady state the interval between consecutive starts / ends and the overlap between
consecutive launches.  usage: overlap_trace.py <dir with *kernel_trace.csv>"""
import csv, glob, sys, statistics as st
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byname = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:60]
    byname.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, v in byname.items():
    d = [e - s for s, e in v]
    print(f"{n:62s} n={len(v):5d} mean {st.mean(d)/1e3:8.2f} us  median {st.median(d)/1e3:8.2f}  min {min(d)/1e3:8.2f}")
for key in [k for k in byname if "stream_scan" in k or "scan_fused" in k]:
    v = byname[key][-200:]
    ds = [b[0] - a[0] for a, b in zip(v, v[1:])]
    de = [b[1] - a[1] for a, b in zip(v, v[1:])]
    ov = [a[1] - b[0] for a, b in zip(v, v[1:])]
    print(key, "last 200: start-to-start median %.2f us, end-to-end median %.2f us, overlap(prev end - next start) median %.2f us" %
          (st.median(ds) / 1e3, st.median(de) / 1e3, st.median(ov) / 1e3))
