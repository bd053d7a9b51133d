#!/bin/bash
# On the GPU box: the plugin path's late ticks under the box's noise, A/B (same box, back to back).
cd tests/host
P=../../mediastreamer2_amd/libmsmi355xfilters.so
nproc; uptime
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d.get('slow_ticks',[])[:3]
print('$1', 'legs',d['legs'],'tickers',d['tickers'],'p50',d['p50_ms'],'p99',d['p99_ms'],'max',d['max_ms'],'late',d['late'],'backlog',d['max_backlog_ms'],'us/leg',d['us_per_leg_tick'], [(s['ms'],s['cpu_ms'],s['flush_ms'],s['nivcsw'],s.get('others_over_8ms')) for s in st])
"; }
for rep in 1 2; do
PLUGIN_BENCH_PACED=1 ./plugin_bench $P 24576 16 600 40 2>/dev/null | show default16
PLUGIN_BENCH_PACED=1 ./plugin_bench $P 24576 8 600 40 2>/dev/null | show tickers8
PLUGIN_BENCH_PACED=1 MSMI355X_NO_EARLY_LAUNCH=1 ./plugin_bench $P 24576 16 600 40 2>/dev/null | show noearly16
PLUGIN_BENCH_PACED=1 MSMI355X_ZERO_COPY=0 ./plugin_bench $P 24576 16 600 40 2>/dev/null | show staged16
PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SHAPE=server ./plugin_bench $P 65536 16 600 40 2>/dev/null | show server16
PLUGIN_BENCH_PACED=1 MALLOC_ARENA_MAX=64 ./plugin_bench $P 24576 16 600 40 2>/dev/null | show arenas16
PLUGIN_BENCH_PACED=1 LD_LIBRARY_PATH=double ./plugin_bench double/libmsmi355xfilters.so 24576 16 600 40 2>/dev/null | show DOUBLE_no_gpu16
done
