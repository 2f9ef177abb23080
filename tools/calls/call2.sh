cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
for d in 0 1 2 3 4 16; do PSH_DBG=$d timeout 120 python tools/lq_ablate.py 126 64; done 2>&1 | grep -v amdgpu.ids > gpurun_out/lq_ablate_2.txt
for w in 64 252; do for d in 0 3; do PSH_DBG=$d timeout 120 python tools/lq_ablate.py $w 64; done; done 2>&1 | grep -v amdgpu.ids >> gpurun_out/lq_ablate_2.txt
PSH_DBG=0 timeout 120 python tools/lq_ablate.py 126 16 2>&1 | grep -v amdgpu.ids >> gpurun_out/lq_ablate_2.txt
PSH_DBG=0 timeout 120 python tools/lq_ablate.py 126 512 2>&1 | grep -v amdgpu.ids >> gpurun_out/lq_ablate_2.txt
unset PSH_LIB
timeout 900 python -m pytest tests -m gpu -x -q -k "long or admitted or batched" 2>&1 | grep -E "passed|failed|error" > gpurun_out/gputests_2.log
