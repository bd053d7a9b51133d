#!/bin/bash
# where a ticker thread's CPU time goes in the plugin path (no perf on the box): plugin_bench's own SIGPROF sampler, named by scripts/walk_profile.py
# usage: scripts/r06_walk_profile.sh [shape] [legs] [tag]
make -C tests/host plugin_bench >/dev/null 2>&1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
SHAPE="${1:-astream default}"; [ "$SHAPE" = std ] && SHAPE=""; LEGS=${2:-32768}; TAG=${3:-a}
PLUGIN_BENCH_SHAPE="$SHAPE" PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SAMPLE=2000 timeout 300 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $LEGS $T 600 20 > gpurun_out/r06_walk_$TAG.json 2> gpurun_out/r06_walk_$TAG.stderr
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_walk_$TAG.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('legs','tickers','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','ticker_graph_walk_ms','ticker_flush_ms','launches_per_tick_and_ticker','flush_rounds_per_tick_and_ticker','fused_legs','recv_streams')})"
{ echo "== shape '$SHAPE' legs $LEGS tickers $T"; python3 scripts/walk_profile.py gpurun_out/r06_walk_$TAG.stderr 70; } > gpurun_out/r06_walk_$TAG.txt
rm -f gpurun_out/r06_walk_$TAG.stderr
head -60 gpurun_out/r06_walk_$TAG.txt
