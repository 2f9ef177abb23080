cd $GRAFT_REPO_ROOT
python tools/batch_sweep.py 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_overlap.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -4
