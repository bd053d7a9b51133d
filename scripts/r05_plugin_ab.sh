#!/bin/bash
# plugin path A/B on the GPU box: speaker-frame slab on / off at two leg counts, paced, 16 tickers (round 5)
set -u
mkdir -p gpurun_out/r05f
make -C tests/host -s plugin_bench >/dev/null 2>&1
P=mediastreamer2_amd/libmsmi355xfilters.so
for L in 32768 49152; do
  for rep in 1 2 3; do
    for ab in pf nopf; do
      E=""; [ $ab = nopf ] && E="MSMI355X_NO_PREFETCH=1"
      env $E PLUGIN_BENCH_PACED=1 tests/host/plugin_bench $P $L 16 600 40 2>/dev/null | tail -1 > gpurun_out/r05f/${ab}_${L}_$rep.json
      python3 -c "
import json; d=json.load(open('gpurun_out/r05f/${ab}_${L}_$rep.json')); print('$ab', $L, $rep, 'p50', d['p50_ms'], 'p99', d['p99_ms'], 'max', d['max_ms'], 'late', d['late'], 'us/leg-tick', d['us_per_leg_tick'], 'walk', d['ticker_graph_walk_ms'], 'flush', d['ticker_flush_ms'], 'from_attach max', d['from_attach']['max_ms'], d['from_attach']['first_ms'][:4])"
    done
  done
done
MS2SHIM_PROFILE=1 PLUGIN_BENCH_PACED=1 tests/host/plugin_bench $P 49152 16 300 40 2>/dev/null | tail -1 > gpurun_out/r05f/walk_49152.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05f/walk_49152.json')); print('walk by id', d['walk_us_per_leg_tick_by_filter_id'], 'flush us/leg', d['ticker_flush_ms']*1e3*16/49152)"
