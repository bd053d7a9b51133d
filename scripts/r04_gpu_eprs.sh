#!/bin/bash
# the conference as MSAudioConference plumbs it (every pin between an in_resampler and an out_resampler) at config[3]'s count, paced
set -u
make -C tests/host -s plugin_bench libms2shim.so >/dev/null 2>&1
for sh in "" "eprs"; do
PLUGIN_BENCH_SHAPE="$sh" PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 16 1000 100 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shape [$sh]', {k:d[k] for k in ('legs','fused_legs','p50_ms','p99_ms','max_ms','ticker_graph_walk_ms','ticker_flush_ms','us_per_leg_tick','launches_per_tick_and_ticker','late_events')})"
done
python -m pytest tests/test_gpu_plugin_fused.py -q -k "config3" 2>&1 | grep -v ms2shim | tail -3
