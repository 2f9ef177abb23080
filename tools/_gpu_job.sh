cd $GRAFT_REPO_ROOT
python tools/batch_sweep.py 2>/dev/null
PSH_LIB=shadowing_amd/lib/libpsh_hip_tuning.so PSH_MQ_I8=2 python tools/batch_sweep.py 2>/dev/null | sed 's/^/i8 /'
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline --no-parity 2>/dev/null | cut -c150-260
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_overlap.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
