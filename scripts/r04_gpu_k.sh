#!/bin/bash
# round 4, GPU pass K4: the slow steps -- the shim's per-filter-id profile AND the plugin's own trace in one run
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04k4_plugin_trace.txt; : > $O
for rep in 1 2 3 4; do
  echo "== rep $rep" | tee -a $O
  MS2SHIM_PROFILE=1 MSMI355X_TRACE_SLOW_MS=3 timeout 600 $PB $PL 32768 16 1000 40 2>/tmp/pb.err >/tmp/pb.json
  grep -a "plugin_bench profile" /tmp/pb.err | tee -a $O
  grep -a -A6 "mi355x leg bank\|mi355x mixer" /tmp/pb.err | grep -a -v "Getting reference\|^$\|Not enough\|^--" | awk '/tick [0-9]+/ { match($0, /tick [0-9]+/); t = substr($0, RSTART + 5, RLENGTH - 5) + 0; keep = (t > 45) } keep' | head -60 | tee -a $O
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms')})
for s in d['slow_ticks']: print('   ',s)" | tee -a $O
done
