#!/bin/bash
# round 4, GPU pass K5: where a thread is when its step runs long (watchdog + backtrace); then volmix parity + rate
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04k5_plugin_stacks.txt; : > $O
for rep in 1 2 3; do
  echo "== rep $rep" | tee -a $O
  PLUGIN_BENCH_STACKS=7 timeout 600 $PB $PL 32768 16 1000 40 2>/tmp/pb.err >/tmp/pb.json
  grep -a -A26 "plugin_bench: a step" /tmp/pb.err | grep -a -v "Getting reference\|^$\|Not enough\|ms2shim-warning" | head -150 | tee -a $O
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms')})" | tee -a $O
done
bash scripts/r04_gpu_l.sh
