#!/bin/bash
# where the attaching threads are while they attach (every filter's preprocess, this plugin's fusing): the sampler of r06_walk_profile.sh over the attach
# usage: scripts/r06_attach_profile.sh [shape] [legs=32768] [tag]
make -C tests/host plugin_bench >/dev/null 2>&1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
SHAPE="${1:-}"; [ "$SHAPE" = std ] && SHAPE=""; LEGS=${2:-32768}; TAG=${3:-a}
PLUGIN_BENCH_SHAPE="$SHAPE" PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SAMPLE=4000 PLUGIN_BENCH_SAMPLE_ATTACH=1 timeout 300 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $LEGS $T 30 1 > gpurun_out/r06_attachprof_$TAG.json 2> gpurun_out/r06_attachprof_$TAG.stderr
A=$(python3 -c "
import json
d=json.loads(open('gpurun_out/r06_attachprof_$TAG.json').read().strip().splitlines()[-1]); print(d['attach_ms_slowest_ticker'])")
{ echo "== the attach (+ one warm-up tick): shape '$SHAPE' legs $LEGS tickers $T; the slowest ticker's attach took $A ms"; python3 scripts/walk_profile.py gpurun_out/r06_attachprof_$TAG.stderr 40; } > gpurun_out/r06_attachprof_$TAG.txt
rm -f gpurun_out/r06_attachprof_$TAG.stderr
head -90 gpurun_out/r06_attachprof_$TAG.txt
