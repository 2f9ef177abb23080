set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batched.py tests/test_gpu_nonfinite.py tests/test_gpu_configs3.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt
for i in 1 2; do timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $O/q512_stages.txt; done
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $O/bench_q512.json 2> $O/bench_q512.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/p/pmc_1 -o p -- python $R/bench.py --steps 2 --warmup 1 --queries 512 --no-cpu-baseline --no-parity > $O/pmc.log 2>&1
python $R/tools/summarize_pmc.py $O/p | grep -E "scan_mq" > $O/pmc_lds.txt; rm -rf $O/p
cd $R
cat $O/q512_stages.txt $O/pmc_lds.txt | cut -c1-200; tail -n 3 $O/tests.txt; cut -c1-260 $O/bench_q512.json
