cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_batched.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
for d in 0 4 8; do echo "PSH_DBG=$d"; PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" | cut -c1-330; done
timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline 2>/dev/null | cut -c150-420
