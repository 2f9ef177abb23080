cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/gputests_start.log
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
for d in 0 1 2 3 4 5 6 7 16; do PSH_DBG=$d timeout 120 python tools/lq_ablate.py 126 64; done > gpurun_out/lq_ablate_start.txt 2>&1
for d in 0 3; do PSH_DBG=$d timeout 120 python tools/lq_ablate.py 252 64; PSH_DBG=$d timeout 120 python tools/lq_ablate.py 64 64; done >> gpurun_out/lq_ablate_start.txt 2>&1
