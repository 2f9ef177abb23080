#!/bin/bash
# Counters of the batched long-window scan (scan_lq_kernel, W = 126, 64 queries): own runs, no trace options.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lqpmc
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/long_batch_probe.py --W 126 --B 64 --steps 2 > $OUT/pmc_$i.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT | grep -E "^==|scan_lq_kernel" > $OUT/lq_pmc_summary.txt 2>&1
cat $OUT/lq_pmc_summary.txt | cut -c1-600
