#!/bin/bash
# A/B of the long-window scan on ONE box: the library in the tree against shadowing_amd/lib/libpsh_prev.so (built from the
# previous commit), the scan kernel alone (tools/long_ablate.py: events around it) and bench.py's per-step figure
for W in 64 126 252; do
  for lib in prev new; do
    if [ $lib = prev ]; then export PSH_LIB=$PWD/shadowing_amd/lib/libpsh_prev.so; else unset PSH_LIB; fi
    echo "== W=$W lib=$lib"
    python tools/long_ablate.py $W 2>&1 | tail -1
    python bench.py --W $W --steps 200 --no-cpu-baseline --no-blocking-api --repeats 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('   bench: %.1f us per step on %d streams, one stream %.1f' % (1e3*j['ms_per_step'], j['streams'], 1e3*j['single_stream_fused']['ms_per_step']))"
  done
done
