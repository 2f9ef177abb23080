#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/rowspmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/bench_forward_topk.py > $OUT/pmc_$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
out=os.environ.get("GRAFT_REPO_ROOT",os.getcwd())+"/gpurun_out/rowspmc"
for path in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if "rows_kernel<1>" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(acc.items()): print(f"{k:26s} n={len(v)} min={min(v):.3e} max={max(v):.3e}")
PY
