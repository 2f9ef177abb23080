#!/bin/bash
# Copy the summaries of the last `gpurun -- bash tools/profile_round.sh` (merged back under gpurun_out/round/) into
# profiles/ under this round's names:  tools/refresh_profiles.sh r01
set -eu
TAG=${1:?round tag, e.g. r01}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/round
cp $S/bench_n1.json $R/profiles/${TAG}_bench_n1.json
cp $S/bench_n1_q512.json $R/profiles/${TAG}_bench_n1_q512.json
cp $S/bench_n1_valu_filter.json $R/profiles/${TAG}_bench_n1_valu_filter.json
cp $S/kernel_stats.csv $R/profiles/${TAG}_rocprofv3_kernel_stats.csv
cp $S/pmc_summary.txt $R/profiles/${TAG}_rocprofv3_pmc_summary.txt
cp $S/hbm_traffic.json $R/profiles/hbm_traffic.json
[ -s $S/bench_foveal.jsonl ] && cp $S/bench_foveal.jsonl $R/profiles/${TAG}_bench_foveal.jsonl
[ -s $S/bench_forward_topk.jsonl ] && cp $S/bench_forward_topk.jsonl $R/profiles/${TAG}_bench_forward_topk.jsonl
[ -s $S/foveal_kernel_stats.csv ] && cp $S/foveal_kernel_stats.csv $R/profiles/${TAG}_foveal_kernel_stats.csv
ls -la $R/profiles
