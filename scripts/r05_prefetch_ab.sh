#!/bin/bash
# On the GPU box: the facades' prefetch distance, A/B on the same box, interleaved (49 152 and 32 768 sending legs, 16 tickers, paced).
cd tests/host
P=../../mediastreamer2_amd/libmsmi355xfilters.so
nproc; uptime
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'legs',d['legs'],'p50',d['p50_ms'],'p99',d['p99_ms'],'max',d['max_ms'],'late',d['late'],'us/leg',d['us_per_leg_tick'],'walk',d['ticker_graph_walk_ms'],'flush',d['ticker_flush_ms'])
"; }
for rep in 1 2 3; do
for n in 32768 49152; do
for a in 1 2 3 4; do
PLUGIN_BENCH_PACED=1 MSMI355X_PREFETCH_AHEAD=$a ./plugin_bench $P $n 16 400 40 2>/dev/null | show ahead$a
done; done; done
uptime
