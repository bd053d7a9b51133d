#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== fused graph leftovers"
timeout 300 python tests/fused_graph.py plain 2>/dev/null | tail -1 | cut -c1-600
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
for rep in 1 2; do
echo "== 32768/16"; timeout 600 $PB $PL 32768 16 600 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','worst_tick','ticker_flush_ms','ticker_graph_walk_ms')})"
echo "== 32768/14"; timeout 600 $PB $PL 32768 14 600 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','worst_tick','ticker_flush_ms','ticker_graph_walk_ms')})"
echo "== 32768/16 malloc tuned"; MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 MALLOC_MMAP_THRESHOLD_=4294967295 timeout 600 $PB $PL 32768 16 600 40 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','worst_tick','ticker_flush_ms','ticker_graph_walk_ms')})"
done
