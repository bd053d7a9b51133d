#!/bin/bash
# where a ticker thread's graph walk goes, by filter id (MS2SHIM_PROFILE), at config[3]'s 32 768 legs on 16 tickers, paced
set -u
mkdir -p gpurun_out
make -C tests/host -s plugin_bench libms2shim.so >/dev/null 2>&1
for i in 1 2; do
PLUGIN_BENCH_PACED=1 MS2SHIM_PROFILE=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 16 300 50 2>gpurun_out/walk_prof_$i.err | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','ticker_graph_walk_ms','ticker_flush_ms','ticker_cpu_ms','us_per_leg_tick')})"
grep "plugin_bench profile" gpurun_out/walk_prof_$i.err | cut -c1-600
done
