cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4ad; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $O/tests.txt
for d in 0 4 8; do echo "PSH_MQ_I8=1 PSH_DBG=$d" >> $O/abl.txt; PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" | cut -c1-330 >> $O/abl.txt; done
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $O/bench_q512.json 2> $O/bench_q512.err
tail -n 25 $O/tests.txt; cat $O/abl.txt; cut -c1-300 $O/bench_q512.json; tail -n 5 $O/bench_q512.err
