#!/bin/bash
# Copy the summaries of the last `gpurun -- bash tools/profile_round.sh` (merged back under gpurun_out/round/) into
# profiles/ under this round's names:  tools/refresh_profiles.sh r02
set -eu
TAG=${1:?round tag, e.g. r02}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/round
cp $S/bench_n1.json $R/profiles/${TAG}_bench_n1.json
cp $S/bench_n1_q512.json $R/profiles/${TAG}_bench_n1_q512.json
cp $S/bench_n1_valu_filter.json $R/profiles/${TAG}_bench_n1_valu_filter.json
cp $S/kernel_stats.csv $R/profiles/${TAG}_rocprofv3_kernel_stats.csv
cp $S/pmc_summary.txt $R/profiles/${TAG}_rocprofv3_pmc_summary.txt
cp $S/hbm_traffic.json $R/profiles/hbm_traffic.json
[ -s $S/q512_pmc.json ] && cp $S/q512_pmc.json $R/profiles/q512_pmc.json
for f in bench_n1_20steps.json bench_n1_fused_one_stream.json kernel_stats_fused_one_stream.csv overlap_trace_summary.txt pmc_mq_summary.txt bench_n1_separate_launches.json bench_n1_forced_exchange.json fused_phase_times.txt bench_foveal.jsonl bench_forward_topk.jsonl \
         foveal_kernel_stats.csv forward_topk_kernel_stats.csv kernel_stats_nofuse.csv foveal_testing_kernel_stats.csv wavelet_kernel_stats.csv \
         embedded_pmc_summary.txt forward_topk_stages.txt q512_stages.txt k_sweep.txt \
         bench_n1_R131072.json fused_skeleton.json mq_ablations.txt ubench_lds_unaligned.txt \
         bench_n1_q512_f16_test.json ubench_mfma_i8.txt q512_kernel_stats.csv mq_clock.txt mq8_phases.txt batch_sweep.txt \
         bench_n1_W64.json bench_n1_W126.json bench_n1_W252.json bench_n1_W126_valu_filter.json blocking_probe.txt ubench_mfma_cover.txt W126_kernel_stats.csv emx_phases.txt long_batch_probe.jsonl long_ablate.txt \
         blocking_times.txt lq_stages.jsonl lq_ablate.txt long_pmc_summary.txt lq_pmc_summary.txt long_sample_ab.txt; do
  [ -s $S/$f ] && cp $S/$f $R/profiles/${TAG}_$f
done
ls -la $R/profiles
