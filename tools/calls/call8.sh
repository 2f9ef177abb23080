cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "long or admitted" 2>&1 | grep -E "passed|failed|error|Error" | head -20 > gpurun_out/gputests_8.log
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
for w in 64 126 252; do
  for skip in 8 0; do
    PSH_STREAM_SKIP=$skip timeout 200 python bench.py --W $w --steps 200 --no-cpu-baseline --no-blocking-api 2>/dev/null | tail -1 > gpurun_out/bench8_W${w}_skip$skip.json
  done
done
