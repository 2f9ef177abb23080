cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/long_batch_probe.py --W 26 30 33 --B 4 16 64 --steps 20 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['data'], 'W', d['W'], 'B', d['B'], 'path', d['path'], 'call_ms', d['call_ms'], 'valu one pass', d['one_pass_valu_ms'], 'same', d['same'])
" > gpurun_out/short_lq.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nonfinite.py tests/test_gpu_admitted_set.py tests/test_gpu_batched.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_20.log
