set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt
timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $O/q512_stages.txt
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $O/bench_q512.json 2> $O/bench_q512.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o t -- python $R/bench.py --steps 10 --warmup 3 --queries 512 --no-cpu-baseline --no-parity > $O/tr.log 2>&1
for f in $(find $O/tr -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $O/q512_kernel_stats.csv; done
rm -rf $O/tr
cat $O/q512_stages.txt | cut -c1-330; cat $O/q512_kernel_stats.csv | cut -c1-150; tail -n 3 $O/tests.txt; cut -c1-260 $O/bench_q512.json
