#!/bin/bash
# the test program's allocator settings (mallopt M_TOP_PAD / M_TRIM_THRESHOLD against glibc's defaults), three interleaved pairs, steady state and first ticks
make -C tests/host plugin_bench >/dev/null 2>&1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
for rep in 1 2 3; do for m in pad default; do
  if [ $m = default ]; then export PLUGIN_BENCH_DEFAULT_MALLOC=1; else unset PLUGIN_BENCH_DEFAULT_MALLOC; fi
  PLUGIN_BENCH_SHAPE="${1:-}" PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so ${2:-32768} $T 300 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', {k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','us_per_leg_tick','ticker_graph_walk_ms')}, d['from_attach']['first_ms'][:4])"
done; done
