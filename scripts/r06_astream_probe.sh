#!/bin/bash
# Round 6: the full-duplex AudioStream shape through the plugin, paced, with and without the per-filter walk profile.
# usage: scripts/r06_astream_probe.sh <tag> [legs=32768] [shape=astream]
tag=${1:-base}; legs=${2:-32768}; shape=${3:-astream}
out=gpurun_out/r06_astream_${tag}.txt
make -C tests/host plugin_bench >/dev/null 2>&1
nproc > $out; uptime >> $out
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
for rep in 1 2; do
PLUGIN_BENCH_SHAPE="$shape" PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $legs $T 250 40 >> $out 2>&1
done
PLUGIN_BENCH_SHAPE="$shape" PLUGIN_BENCH_PACED=1 MS2SHIM_PROFILE=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $legs $T 250 40 >> $out 2>&1
PLUGIN_BENCH_SHAPE="$shape" tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $legs $T 250 40 >> $out 2>&1
uptime >> $out
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("{"):
        d=json.loads(l)
        print({k:d[k] for k in ("paced","legs","tickers","p50_ms","p99_ms","max_ms","late","us_per_leg_tick","ticker_flush_ms","ticker_graph_walk_ms","launches_per_tick_and_ticker","flush_rounds_per_tick_and_ticker","fused_legs","late_events","walk_us_per_leg_tick_by_filter_id")}, d["from_attach"])
PY
