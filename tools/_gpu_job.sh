set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4r; mkdir -p $O
bash tools/mq_clock.sh > $O/mq_clock.txt 2>&1
cat $O/mq_clock.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $O/all_tests.txt
timeout 300 python tools/k_sweep.py 2>/dev/null | grep "us per call" > $O/k_sweep.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -n 5 $O/all_tests.txt; cat $O/k_sweep.txt; tail -n 2 $O/smoke.txt
