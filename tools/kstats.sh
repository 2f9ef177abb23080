#!/bin/bash
# rocprofv3 per-kernel statistics of one command on the GPU box:  tools/kstats.sh <name> <command...>
# -> gpurun_out/kstats/<name>.csv (+ the command's own output in <name>.log)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/kstats
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kstats_$NAME
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_$NAME -o k -- "$@" > $OUT/$NAME.log 2>&1
for f in $(find /tmp/kstats_$NAME -name "*kernel_stats.csv"); do cp $f $OUT/$NAME.csv; done
head -12 $OUT/$NAME.csv
