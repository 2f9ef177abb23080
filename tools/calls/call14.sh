cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so timeout 300 python tools/smooth_dbg.py 2>&1 | grep -v amdgpu.ids > gpurun_out/smooth_dbg.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_14.log
