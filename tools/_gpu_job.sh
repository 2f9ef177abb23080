set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f; mkdir -p $O

timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_overlap.py tests/test_gpu_configs3.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/tests.txt
for i in 1 2; do timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $O/q512_stages.txt; done
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $O/bench_q512.json 2> $O/bench_q512.err
export PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so
for d in 0 4 8 12; do echo "DBG=$d" >> $O/mq_dbg.txt; PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $O/mq_dbg.txt; done
unset PSH_LIB
cat $O/q512_stages.txt $O/mq_dbg.txt; tail -n 6 $O/tests.txt; cat $O/bench_q512.json | cut -c1-300
