#!/bin/bash
# cross-hub batching, round 5: (1) the device's side alone (scripts/r05_cross_hub.py), (2) the plugin's 16 hubs as they launch today
# under rocprofv3 (kernel time per tick from the trace's stats)
set -u
mkdir -p gpurun_out/r05x
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for L in 32768 65536; do
  timeout 300 python3 scripts/r05_cross_hub.py $L 16 2>/dev/null | tail -1 | tee gpurun_out/r05x/device_side_$L.json
done
make -C tests/host -s plugin_bench >/dev/null 2>&1
for L in 32768 65536; do
  rm -rf gpurun_out/r05x/prof_$L
  PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_CLEAN_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05x/prof_$L -o plugin -- tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $L 16 400 40 > gpurun_out/r05x/plugin_$L.json 2> gpurun_out/r05x/plugin_$L.err
  f=$(find gpurun_out/r05x/prof_$L -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r05x/plugin_${L}_kernel_stats.csv && head -8 $f | cut -c1-200
  find gpurun_out/r05x/prof_$L -name "*kernel_trace.csv" -delete
  tail -1 gpurun_out/r05x/plugin_$L.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print($L, 'p50', d['p50_ms'], 'p99', d['p99_ms'], 'late', d['late'], 'flush', d['ticker_flush_ms'], 'walk', d['ticker_graph_walk_ms'])"
done
