#!/bin/bash
# does the walk's cost follow the TLB?  the standard leg shape at 3072 legs per ticker with glibc's malloc on transparent huge pages
make -C tests/host plugin_bench >/dev/null 2>&1
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
ldd --version | head -1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
for rep in 1 2 3; do
for tun in "" "glibc.malloc.hugetlb=1"; do
  GLIBC_TUNABLES="$tun" PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so ${1:-49152} $T 250 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tun' or 'default', {k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','ticker_flush_ms','ticker_graph_walk_ms','minflt_per_tick_and_ticker')})"
done; done
