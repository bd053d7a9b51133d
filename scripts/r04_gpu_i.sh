#!/bin/bash
# round 4, GPU pass I: what the plugin path's rare ~10 ms ticks are (per-tick CPU time / context switches / page faults of the slowest thread)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04i_plugin_diag.jsonl; : > $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
run() { echo "== $*"; echo "{\"run\": \"$*\"}" >> $O; timeout 900 "$@" 2>/dev/null | tail -1 | tee -a $O | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('legs','tickers','p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms','ticker_cpu_ms','minflt_per_tick_and_ticker','nvcsw_per_tick_and_ticker','nivcsw_per_tick_and_ticker','max_backlog_ms')})
for s in d['slow_ticks']: print('   ',s)"; }
run $PB $PL 32768 16 1000 40
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
run env GPU_MAX_HW_QUEUES=16 $PB $PL 32768 16 1000 40
run env MALLOC_ARENA_MAX=64 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=268435456 MALLOC_MMAP_THRESHOLD_=1073741824 $PB $PL 32768 16 1000 40
run $PB $PL 32768 8 600 40
run $PB $PL 32768 32 600 40
run env MSMI355X_NO_EARLY_LAUNCH=1 $PB $PL 32768 16 600 40
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
