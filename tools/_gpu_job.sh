cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4ag; mkdir -p $O
L=$R/shadowing_amd/lib
timeout 1500 python -m pytest tests/test_gpu_embedded.py tests/test_gpu_nonfinite.py tests/test_gpu_predict.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt
for i in 1 2 3; do
for v in prev new; do
  if [ $v = prev ]; then export PSH_LIB=$L/libpsh_hip_prev.so; else unset PSH_LIB; fi
  timeout 300 python tools/bench_foveal.py --which tutorial testing --steps 30 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print('$v', j['workload'][:8], j['ms_per_call'], j['stages_ms']['scan_ms'])
" >> $O/ab.txt
done; done
tail -n 4 $O/tests.txt; cat $O/ab.txt
