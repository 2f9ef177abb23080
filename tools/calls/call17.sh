cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/small_long_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_long_batch.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "long" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_17.log
