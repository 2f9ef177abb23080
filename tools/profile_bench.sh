#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM PMC passes.
# Everything lands under gpurun_out/; summaries worth keeping are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python bench.py --steps ${STEPS:-200} --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o scan -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python $R/tools/summarize_pmc.py $OUT
