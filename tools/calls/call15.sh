cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_configs3.py -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_15.log
timeout 300 python bench.py --gpus 1 --force-sharded --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('forced exchange', d['ms_per_step'], d.get('rccl_world_size'))" >> gpurun_out/gputests_15.log 2>&1
