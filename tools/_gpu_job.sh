set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_nonfinite.py tests/test_gpu_predict.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/new_tests.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/all_tests.txt
export PSH_LIB=$GRAFT_REPO_ROOT/shadowing_amd/lib/libpsh_hip_tuning.so
for d in 0 1 2 3 4 8 12 0; do echo "DBG=$d" >> $O/mq_dbg.txt; PSH_DBG=$d timeout 120 python tools/q512_stages.py 2>/dev/null | grep "^{" >> $O/mq_dbg.txt; done
timeout 300 python tools/fused_skeleton.py > $O/fused_skeleton.json 2> $O/fused_skeleton.err
unset PSH_LIB
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fov -o f -- python $GRAFT_REPO_ROOT/tools/fov_prof.py 0 > $O/fov_prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $O/prof_fov -name "*kernel_stats.csv"); do grep "Name\|psh::" $f > $O/foveal_kernel_stats.csv; done
rm -rf $O/prof_fov
tail -n 8 $O/new_tests.txt $O/all_tests.txt; cat $O/mq_dbg.txt $O/fused_skeleton.json $O/foveal_kernel_stats.csv
