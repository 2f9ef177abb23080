cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "long" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_10.log
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
(for skip in 8 0 8 0; do echo "PSH_STREAM_SKIP=$skip"; PSH_STREAM_SKIP=$skip timeout 300 python tools/long_batch_probe.py --W 64 126 252 --B 1 2 3 --steps 50 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  W', d['W'], 'B', d['B'], 'call_ms', d['call_ms'], 'same', d['same'])
"; done) > gpurun_out/lone_stream_ab.txt 2>&1
