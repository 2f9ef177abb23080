cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4ab; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $O/tests.txt
tail -n 25 $O/tests.txt
