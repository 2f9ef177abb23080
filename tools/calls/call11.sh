cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_11.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/gputests_11.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_driver_form.json
