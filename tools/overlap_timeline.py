"""From a rocprofv3 --kernel-trace csv of tools/long_probe.py (TRACE=1): split the stream_scan launches into timed regions
(gaps > 1 ms), print per region the start-to-start interval statistics and, for the slowest and the fastest region, a
stretch of the kernel timeline (queue, kernel, start, end relative to the stretch)."""
import csv, glob, sys, statistics as st
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows if "psh::stream" in r["Kernel_Name"]]
ks.sort()
scans = [x for x in ks if "stream_scan" in x[2]]
regions, cur = [], [scans[0]]
for a, b in zip(scans, scans[1:]):
    if b[0] - a[1] > 1_000_000:
        regions.append(cur); cur = []
    cur.append(b)
regions.append(cur)
stats = []
for i, rg in enumerate(regions):
    if len(rg) < 100:
        continue
    ends = sorted(x[1] for x in rg)
    iv = [(b - a) / 1e3 for a, b in zip(ends, ends[1:])]
    per = (ends[-1] - ends[0]) / (len(ends) - 1) / 1e3
    conc = []
    for x in rg[10:-10]:
        conc.append(sum(1 for y in rg if y[0] < x[1] and y[1] > x[0]) - 1)
    stats.append((per, i))
    print(f"region {i}: {len(rg)} scans, end-to-end {per:6.2f} us/step, interval median {st.median(iv):6.2f} p90 {sorted(iv)[int(0.9 * len(iv))]:6.2f}; "
          f"scan duration median {st.median((x[1] - x[0]) / 1e3 for x in rg):6.1f}; other scans overlapping a scan: median {st.median(conc)}")
def show(i, label):
    rg = regions[i]
    t0, t1 = rg[150][0], rg[150][0] + 700_000
    print(f"--- {label} region {i}: kernels between +0 and +700 us of its 150th scan")
    for s, e, n, qd in ks:
        if e > t0 and s < t1:
            nm = "P" if "sample" in n else ("S" if "scan" in n else "R")
            print(f"   q{qd:>3s} {nm}  {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:6.1f})")
stats.sort()
show(stats[0][1], "fastest"); show(stats[-1][1], "slowest")
