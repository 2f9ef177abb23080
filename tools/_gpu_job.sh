cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_batched.py tests/test_gpu_overlap.py -q -m gpu 2>&1 | tail -15
