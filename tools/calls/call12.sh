cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "long or admitted or smooth or nonfinite" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_12.log
(timeout 300 python tools/long_batch_probe.py --walk --W 20 64 126 --B 1 4 16 --steps 20 2>/dev/null | grep "^{"; timeout 300 python tools/long_batch_probe.py --W 64 126 252 --B 1 3 4 64 --steps 20 2>/dev/null | grep "^{") | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['data'], 'W', d['W'], 'B', d['B'], 'path', d['path'], 'call_ms', d['call_ms'], 'same', d['same'])
" > gpurun_out/probe_12.txt 2>&1
timeout 300 python tools/lq_stages.py 126 2>/dev/null | grep "^{" | cut -c1-230 >> gpurun_out/probe_12.txt
for w in 64 126 252; do timeout 200 python bench.py --W $w --steps 200 --no-cpu-baseline --no-blocking-api 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($w, d['ms_per_step'], d['ms_per_step_repeats']['median'], d['parity_rotating_queries']['ok'])"; done >> gpurun_out/probe_12.txt 2>&1
