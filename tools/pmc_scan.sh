#!/bin/bash
# PMC passes over a short bench run (queries = $1), summarised per kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
Q=${1:-1}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_q${Q}_$i -o p -- python $R/bench.py --steps 5 --warmup 2 --queries $Q --no-cpu-baseline --no-parity > $OUT/pmc_q${Q}_$i.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT
