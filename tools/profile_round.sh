#!/bin/bash
# One call on the GPU box: PMC passes (own runs, no trace options) -> HBM traffic json, then the
# bench line (which reads that json), then rocprofv3 kernel stats of the same bench command.
# Results under gpurun_out/round/; the summaries are copied to profiles/ afterwards (tools/refresh_profiles.sh rNN).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$i -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $OUT/pmc_$i.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT --traffic $OUT/hbm_traffic.json > $OUT/pmc_summary.txt 2>&1
cp $OUT/hbm_traffic.json $R/profiles/hbm_traffic.json 2>/dev/null
cd $R
# the headline line: configs[1], independent steps on 3 streams as the overlap-friendly launches (with the CPU baselines)
timeout 600 python bench.py --steps ${STEPS:-1000} --warmup 20 > $OUT/bench_n1.json 2> $OUT/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_20steps.json 2>> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o scan -- python $R/bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
python $R/tools/overlap_trace.py $OUT/prof_stats > $OUT/overlap_trace_summary.txt 2>&1
# the same workload on ONE stream: the fused single launch (what an isolated caller gets; r02's headline)
cd $R
timeout 300 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --streams 1 > $OUT/bench_n1_fused_one_stream.json 2>> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_fused -o scan -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --streams 1 > $OUT/bench_prof_fused.log 2>&1
for f in $(find $OUT/prof_stats_fused -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_fused_one_stream.csv; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_nofuse -o scan -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fuse > $OUT/bench_prof_nofuse.log 2>&1
for f in $(find $OUT/prof_stats_nofuse -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_nofuse.csv; done
cd $R
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fuse > $OUT/bench_n1_separate_launches.json 2>> $OUT/bench.err
bash tools/pmc_mq.sh > $OUT/pmc_mq_summary.txt 2>&1          # (writes profiles/q512_pmc.json, which the next line reads)
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $OUT/bench_n1_q512.json 2>> $OUT/bench.err
cp $R/profiles/q512_pmc.json $OUT/q512_pmc.json 2>/dev/null
timeout 300 python bench.py --filter valu --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_n1_valu_filter.json 2>> $OUT/bench.err
timeout 300 python bench.py --gpus 1 --force-sharded --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_n1_forced_exchange.json 2>> $OUT/bench.err
timeout 300 python tools/fused_times.py > $OUT/fused_phase_times.txt 2>> $OUT/bench.err
# the widened rows of the scope table: the reference's Foveal workloads, configs[4] (wavelet), forward_topk
timeout 600 python tools/bench_foveal.py --steps 20 --generic --which tutorial testing wavelet 2>> $OUT/bench.err | grep "^{" > $OUT/bench_foveal.jsonl
timeout 300 python tools/bench_forward_topk.py 2>> $OUT/bench.err | grep "^{" > $OUT/bench_forward_topk.jsonl
timeout 300 python tools/k_sweep.py 2>> $OUT/bench.err | grep "us per call" > $OUT/k_sweep.txt
timeout 300 python tools/rows_stages.py 2>> $OUT/bench.err | grep "^[0-9]" > $OUT/forward_topk_stages.txt
timeout 300 python tools/q512_stages.py 2>> $OUT/bench.err | grep "^{" > $OUT/q512_stages.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_foveal -o fov -- python $R/tools/bench_foveal.py --steps 20 --which tutorial testing > $OUT/bench_foveal_prof.log 2>&1
for f in $(find $OUT/prof_foveal -name "*kernel_stats.csv"); do head -8 $f > $OUT/foveal_kernel_stats.csv; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rows -o rows -- python $R/tools/bench_forward_topk.py > $OUT/bench_rows_prof.log 2>&1
for f in $(find $OUT/prof_rows -name "*kernel_stats.csv"); do grep "Name\|rows_kernel\|threshold\|select" $f > $OUT/forward_topk_kernel_stats.csv; done
# the embedded scans alone: the reference's timed workload (prefix-sum scan), configs[4] (matrix cores): kernel stats + counters
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fovt -o f -- python $R/tools/fov_prof.py 0 > $OUT/fov_prof.log 2>&1
for f in $(find $OUT/prof_fovt -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/foveal_testing_kernel_stats.csv; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_emx -o f -- python $R/tools/emx_prof.py 16 > $OUT/emx_prof.log 2>&1
for f in $(find $OUT/prof_emx -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/wavelet_kernel_stats.csv; done
j=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  j=$((j+1))
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/emb/pmc_f$j -o p -- python $R/tools/fov_prof.py 0 > $OUT/pmc_f$j.log 2>&1
  timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/emb/pmc_w$j -o p -- python $R/tools/emx_prof.py 16 > $OUT/pmc_w$j.log 2>&1
done
python $R/tools/summarize_pmc.py $OUT/emb | grep -E "^==|embed_px_kernel<true, 1|embed_mx_kernel<true, 1" > $OUT/embedded_pmc_summary.txt 2>&1
cd $R
# round 4: the headline on an ensemble EIGHT times the memory-side cache (2 GiB: the GB/s do not come from the 256 MB MALL),
# the fused launch's skeleton (what one stream's step cannot go below), the batched scan's ablations, unaligned LDS reads
timeout 600 python bench.py --steps 500 --warmup 20 --rows-per-gpu 131072 --no-cpu-baseline > $OUT/bench_n1_R131072.json 2>> $OUT/bench.err
timeout 300 python tools/fused_skeleton.py > $OUT/fused_skeleton.json 2>> $OUT/bench.err
python -m shadowing_amd._build --tuning > /dev/null 2>&1
# (the 8-bit rejection test -- scan_mq8_kernel, the default -- and the f16 one it replaced, side by side on this box)
for i8 in 1 0; do for d in 0 4 8; do echo "PSH_MQ_I8=$i8 (1 scan_mq8_kernel, 0 scan_mq_kernel) PSH_DBG=$d (0 the kernel; 4 no survivor handling; 8 MFMAs only: no epilogue)" >> $OUT/mq_ablations.txt; PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so PSH_MQ_I8=$i8 PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $OUT/mq_ablations.txt; done; done
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --mq-f16 --no-cpu-baseline > $OUT/bench_n1_q512_f16_test.json 2>> $OUT/bench.err
hipcc --offload-arch=gfx950 -O2 -o /tmp/u8 tools/ubench_mfma_i8.hip 2>/dev/null && /tmp/u8 > $OUT/ubench_mfma_i8.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q512 -o q -- python $R/bench.py --steps 10 --warmup 3 --queries 512 --no-cpu-baseline --no-parity > $OUT/bench_prof_q512.log 2>&1
for f in $(find $OUT/prof_q512 -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/q512_kernel_stats.csv; done
cd $R
(for d in 0 4 8; do echo "PSH_DBG=$d"; PSH_DBG=$d python tools/mq8_phases.py 2>/dev/null; done) > $OUT/mq8_phases.txt
timeout 300 python tools/batch_sweep.py 2>/dev/null > $OUT/batch_sweep.txt
bash tools/mq_clock.sh > $OUT/mq_clock.txt 2>&1
PSH_MQ_I8=0 bash tools/mq_clock.sh >> $OUT/mq_clock.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubl tools/ubench_lds_unaligned.hip 2>/dev/null && /tmp/ubl > $OUT/ubench_lds_unaligned.txt 2>&1
tail -c 600 $OUT/pmc_summary.txt; head -8 $OUT/kernel_stats.csv; cat $OUT/bench_n1.json
# round 5: long Identity windows on the matrix cores (stream_scan_long_kernel) beside the vector-ALU filter they replaced, the
# blocking shadow() with and without admission hints, matrix-core cover of vector work (why configs[2] is where it is)
cd $R
for W in 64 126 252; do timeout 300 python bench.py --W $W --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_n1_W$W.json 2>> $OUT/bench.err; done
timeout 300 python bench.py --W 126 --steps 100 --warmup 10 --no-cpu-baseline --filter valu > $OUT/bench_n1_W126_valu_filter.json 2>> $OUT/bench.err
timeout 300 python tools/blocking_probe.py 2>> $OUT/bench.err | grep -v amdgpu.ids > $OUT/blocking_probe.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/ucover tools/ubench_mfma_cover2.hip 2>/dev/null && /tmp/ucover > $OUT/ubench_mfma_cover.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_W126 -o w -- python $R/bench.py --W 126 --steps 200 --warmup 20 --no-cpu-baseline --no-parity > $OUT/bench_prof_W126.log 2>&1
for f in $(find $OUT/prof_W126 -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/W126_kernel_stats.csv; done
cd $R
# round 5, later: the wavelet scan's phases per wave (tuning build), batches with windows the batched bands do not reach (the
# loop of matrix-core steps inside the call beside the one-pass vector-ALU filter; --walk: price levels, the bounds' worst case),
# the long-window scan with parts switched off
PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so timeout 300 python tools/emx_phases.py 2>/dev/null | grep -v amdgpu.ids > $OUT/emx_phases.txt
(timeout 600 python tools/long_batch_probe.py --W 30 64 126 252 --B 1 2 3 4 16 64 2>/dev/null | grep "^{"; timeout 300 python tools/long_batch_probe.py --walk --W 20 64 126 --B 1 4 2>/dev/null | grep "^{") > $OUT/long_batch_probe.jsonl
(for W in 64 126 252; do for d in 0 4 68 12; do PSH_DBG=$d PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so python tools/long_ablate.py $W 2>/dev/null | tail -1; done; done) > $OUT/long_ablate.txt
# round 6: the blocking shadow() call as one library call (psh_shadow_blocking: the library's own split of a call), the batched
# long-window scan (stage times, ablations: PSH_DBG 1 no MFMAs, 2 no tests, 4 no survivor handling), the long-window scan's counters
timeout 300 python tools/blocking_times.py 2>> $OUT/bench.err | grep -v amdgpu.ids > $OUT/blocking_times.txt
timeout 600 python tools/lq_stages.py 64,126,252 2>> $OUT/bench.err | grep "^{" > $OUT/lq_stages.jsonl
# (PSH_DBG: 1 no MFMA chains, 2 no tests, 3 neither, 4 no survivor handling, 8 survivors queued but not verified, 32 the min trees alone, 16 counts the survivors)
PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so timeout 600 python tools/lq_ablate2.py 126:64:0,1,2,3,4,8,32,16 64:64:0,2,3 252:64:0,2,3 126:4:0 126:16:0 126:512:0 64:512:0 252:512:0 2>/dev/null | grep "^W=" > $OUT/lq_ablate.txt
bash tools/pmc_lq.sh > /dev/null 2>&1; cp $R/gpurun_out/lqpmc/lq_pmc_summary.txt $OUT/lq_pmc_summary.txt 2>/dev/null
# the long-window step with its sample as exact chains (PSH_STREAM_SKIP=8, tuning build) and on the matrix cores, same box
(for W in 64 126 252; do for skip in 8 0; do echo "W=$W PSH_STREAM_SKIP=$skip (8: the exact-chain sample; 0: stream_sample_long_kernel)"; PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so PSH_STREAM_SKIP=$skip timeout 200 python bench.py --W $W --steps 200 --warmup 20 --no-cpu-baseline --no-blocking-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'repeats', d.get('ms_per_step_repeats',{}).get('median'), d.get('ms_per_step_repeats',{}).get('min'), 'parity', d.get('parity_rotating_queries',{}).get('ok'))"; done; done) > $OUT/long_sample_ab.txt 2>&1
bash tools/pmc_long.sh > /dev/null 2>&1; cp $R/gpurun_out/longpmc/long_pmc_summary.txt $OUT/long_pmc_summary.txt 2>/dev/null
