#!/bin/bash
# The batched scan (512 queries) against the MATRIX CORES' SUSTAINED RATE on this part: shader cycles (PMC) and duration (kernel
# trace) of scan_mq8_kernel (PSH_MQ_I8=0: scan_mq_kernel, the f16 test) as it is and with its epilogue removed (tuning build, PSH_DBG=8: MFMAs, fragment reads and the
# per-segment work only) -> clock = cycles / duration, MFMAs per second.  The clock FALLS as the matrix cores get busier: the
# part is power-limited on this instruction long before its 2.5 PFLOP/s.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mqclock
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python -m shadowing_amd._build --tuning > /dev/null 2>&1
export PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so
for d in 0 8; do
  PSH_DBG=$d timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_d$d/pmc_1 -o p -- python $R/bench.py --steps 6 --warmup 2 --queries 512 --no-cpu-baseline --no-parity > $OUT/pmc_$d.log 2>&1
  PSH_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d$d -o t -- python $R/bench.py --steps 6 --warmup 2 --queries 512 --no-cpu-baseline --no-parity > $OUT/trace_$d.log 2>&1
done
python - <<PY
import csv, glob, collections
import os
f16 = os.environ.get("PSH_MQ_I8", "1") == "0"                   # (PSH_MQ_I8=0 tools/mq_clock.sh: the f16 test of rounds 2-4)
for d in (0, 8):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_d%d/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan_mq" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = None
    for f in glob.glob("$OUT/trace_d%d/**/*kernel_stats.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            if "scan_mq" in row["Name"]: dur = float(row["AverageNs"]) * 1e-9
    cyc = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8.0            # per XCD
    busy = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0   # per SIMD
    n = sum(acc["SQ_INSTS_MFMA"]) / len(acc["SQ_INSTS_MFMA"])
    print("PSH_DBG=%d (%s): %.3f ms per launch, %.2fe6 shader cycles per XCD -> %.2f GHz; matrix cores busy %.2fe6 cycles per SIMD = %.0f %%; %.3g MFMAs -> %.2fe10 MFMA/s = %.2f P(FL)OP/s issued (%s)" %
          (d, "the kernel" if d == 0 else "MFMAs only: no epilogue", dur * 1e3, cyc / 1e6, cyc / dur / 1e9, busy / 1e6, 100 * busy / cyc, n, n / dur / 1e10,
           n * (32768 if f16 else 65536) / dur / 1e15, "v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_i32_32x32x32_i8, 8-bit operations"))
PY
