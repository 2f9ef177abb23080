#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export PSH_LIB=$R/shadowing_amd/lib/libpsh_hip_tuning.so
run() { echo "== $*"; env "$@" python tools/two_stream_probe.py overlap 2>&1 | grep "streams=[234]" | awk '{print $2, $4}' | paste -sd' '; }
run X=0
run PSH_STREAM_RGRID=1
run PSH_STREAM_RGRID=3
run PSH_STREAM_UNITS=1024
run PSH_STREAM_UNITS=1024 PSH_STREAM_PGRID=4
run PSH_STREAM_PGRID=4
run PSH_STREAM_PGRID=1
