cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
PSH_LIB=$PWD/shadowing_amd/lib/libpsh_hip_tuning.so timeout 300 python tools/smooth_dbg.py 2>&1 | grep -v amdgpu.ids > gpurun_out/smooth_dbg.txt
