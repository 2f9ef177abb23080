cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/lq_stages.py 64,126,252 2>/dev/null | grep "^{" | cut -c1-200 > gpurun_out/lq_stages_16.txt
timeout 300 python tools/long_batch_probe.py --W 64 126 252 --B 4 5 16 64 --steps 20 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['data'], 'W', d['W'], 'B', d['B'], 'path', d['path'], 'call_ms', d['call_ms'], 'same', d['same'])
" >> gpurun_out/lq_stages_16.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "long or admitted" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_16.log
