cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so
timeout 600 python tools/lq_ablate2.py 126:64:0,1,2,3,4,8,32,33,12 64:64:0,2,4,8,32 252:64:0,2,4,8,32 2>&1 | grep -v amdgpu.ids > gpurun_out/lq_ablate_3.txt
unset PSH_LIB
bash tools/pmc_lq.sh > gpurun_out/lq_pmc_3.txt 2>&1
