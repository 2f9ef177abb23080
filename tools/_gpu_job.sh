cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4am; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | tail -3 > $O/tests.txt
for v in prev new prev new; do
  if [ $v = prev ]; then export PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_prev.so; else unset PSH_LIB; fi
  echo $v >> $O/ab.txt
  timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" | cut -c1-260 >> $O/ab.txt
  timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline --no-parity 2>/dev/null | cut -c150-250 >> $O/ab.txt
done
cat $O/tests.txt $O/ab.txt
