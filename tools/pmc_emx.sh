#!/bin/bash
# The counters of the wavelet scan (configs[4], 16 queries) alone, three separate --pmc passes:  gpurun -- bash tools/pmc_emx.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/emx
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/emx_prof.py 16 2>&1 | tail -2 > $OUT/emx_prof.log
j=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  j=$((j+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_w$j -o p -- python $R/tools/emx_prof.py 16 > $OUT/pmc_w$j.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT | grep -E "^==|embed_mx_kernel<true, 1" > $OUT/pmc_summary.txt 2>&1
cat $OUT/emx_prof.log; cut -c1-150 $OUT/pmc_summary.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_emx -o f -- python $R/tools/emx_prof.py 16 > $OUT/emx_prof_trace.log 2>&1
for f in $(find $OUT/prof_emx -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/wavelet_kernel_stats.csv; done
cat $OUT/wavelet_kernel_stats.csv | cut -c1-160
python $R/tools/emx_phases.py 2>&1 | tail -4 > $OUT/emx_phases.txt; cat $OUT/emx_phases.txt
