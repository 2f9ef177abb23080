cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
(PSH_LQ_FORM=2 timeout 300 python tools/lq_ablate2.py 126:64:0 64:64:0 126:16:0 126:512:0; echo third; timeout 600 python tools/lq_ablate2.py 126:64:0,1,2,4,8,16 64:64:0,2 126:16:0 126:4:0 126:512:0 64:512:0; PSH_LQ_FORM=3 timeout 300 python tools/lq_ablate2.py 252:64:0 200:64:0) 2>&1 | grep -v amdgpu.ids > gpurun_out/lq_ablate_9.txt
unset PSH_LIB
timeout 900 python -m pytest tests -m gpu -x -q -k "long or admitted" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/gputests_9.log
