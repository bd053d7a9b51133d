#!/bin/bash
# round 4: where a ticker thread is when its step runs long -- tests/host/plugin_bench with a watchdog that prints the stack of a
# thread whose step has been running for 7 ms (PLUGIN_BENCH_STACKS), the runtime's per-filter-id profile of the slowest step
# (MS2SHIM_PROFILE) and the plugin's own call-by-call trace of a slow enqueue (MSMI355X_TRACE_SLOW_MS); MSMI355X_COPY=hip /
# MSMI355X_ZERO_COPY=0 bring the runtime's copies back, with which the ~13 ms steps show (profiles/r04_plugin_{stacks,trace}.txt)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04k_plugin_stacks.txt; : > $O
for mode in "MSMI355X_ZERO_COPY=1" "MSMI355X_ZERO_COPY=0 MSMI355X_COPY=hip"; do
  echo "== $mode" | tee -a $O
  env $mode PLUGIN_BENCH_STACKS=7 MS2SHIM_PROFILE=1 MSMI355X_TRACE_SLOW_MS=5 timeout 600 $PB $PL 32768 16 1000 40 2>/tmp/pb.err >/tmp/pb.json
  grep -a "plugin_bench profile" /tmp/pb.err | tee -a $O
  grep -a -A26 "plugin_bench: a step" /tmp/pb.err | grep -a -v "Getting reference\|^$\|Not enough\|ms2shim-warning" | head -120 | tee -a $O
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms')})
for s in d['slow_ticks']: print('   ',s)" | tee -a $O
done
