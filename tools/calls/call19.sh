cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nonfinite.py tests/test_gpu_admitted_set.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_19.log
