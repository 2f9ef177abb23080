cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round; mkdir -p $OUT
cp $R/gpurun_out/round_q512_pmc.json $R/profiles/q512_pmc.json 2>/dev/null
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --queries 512 --no-cpu-baseline > $OUT/bench_n1_q512_try$i.json 2>> $OUT/bench.err; done
timeout 600 python tools/bench_foveal.py --steps 20 --generic --which tutorial testing wavelet 2>> $OUT/bench.err | grep "^{" > $OUT/bench_foveal.jsonl
for i in 1 2 3; do python -c "
import json; j=json.loads(open('$OUT/bench_n1_q512_try$i.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['avg_launch_ms'])"; done
python -c "
import json
for ln in open('$OUT/bench_foveal.jsonl'):
    j=json.loads(ln); print(j['workload'][:10], j['ms_per_call'], j['stages_ms']['scan_ms'])"
