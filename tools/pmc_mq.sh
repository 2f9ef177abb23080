#!/bin/bash
# PMC passes over the batched-query bench (queries = 512), summarised per kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$i -o p -- python $R/bench.py --steps 2 --warmup 1 --queries 512 --no-cpu-baseline --no-parity > $OUT/pmc_$i.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT --mq $OUT/q512_pmc.json | grep -E "scan_mq|boot_mq|^==|wrote"
cp $OUT/q512_pmc.json $R/profiles/q512_pmc.json 2>/dev/null
