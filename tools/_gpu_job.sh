cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/pmc_mq.sh > $OUT/pmc_mq_summary.txt 2>&1
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $OUT/bench_n1_q512.json 2>> $OUT/bench.err
cp $R/profiles/q512_pmc.json $OUT/q512_pmc.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --mq-f16 --no-cpu-baseline > $OUT/bench_n1_q512_f16_test.json 2>> $OUT/bench.err
timeout 300 python tools/q512_stages.py 2>> $OUT/bench.err | grep "^{" > $OUT/q512_stages.txt
rm -f $OUT/mq_ablations.txt
for i8 in 1 0; do for d in 0 4 8; do echo "PSH_MQ_I8=$i8 (1 scan_mq8_kernel, 0 scan_mq_kernel) PSH_DBG=$d (0 the kernel; 4 no survivor handling; 8 MFMAs only: no epilogue)" >> $OUT/mq_ablations.txt; PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so PSH_MQ_I8=$i8 PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $OUT/mq_ablations.txt; done; done
cd /tmp
rm -rf $OUT/prof_q512
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q512 -o q -- python $R/bench.py --steps 10 --warmup 3 --queries 512 --no-cpu-baseline --no-parity > $OUT/bench_prof_q512.log 2>&1
for f in $(find $OUT/prof_q512 -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $OUT/q512_kernel_stats.csv; done
cd $R
(for d in 0 4 8; do echo "PSH_DBG=$d"; PSH_DBG=$d python tools/mq8_phases.py 2>/dev/null; done) > $OUT/mq8_phases.txt
timeout 300 python tools/batch_sweep.py 2>/dev/null > $OUT/batch_sweep.txt
cut -c1-300 $OUT/bench_n1_q512.json; cat $OUT/q512_kernel_stats.csv | cut -c1-120; cat $OUT/batch_sweep.txt
