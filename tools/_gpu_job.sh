cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4af; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_batched.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/tests.txt
timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" | cut -c1-330 >> $O/abl.txt
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $O/bench_q512.json 2> $O/bench_q512.err
tail -n 40 $O/tests.txt; cat $O/abl.txt; cut -c1-300 $O/bench_q512.json
