cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default20.json 2> $O/bench.err
tail -n 4 $O/tests.txt; tail -n 1 $O/smoke.txt; cut -c1-330 $O/bench_default20.json
