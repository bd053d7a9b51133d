#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of tests/host/plugin_bench (the drop-in plugin's launches) per leg shape, 32 768 legs on 16 tickers, 200 paced ticks.
set -u
OUT=gpurun_out/r06_plugin_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cd tests/host
for sh in default astream astream_default; do
	S="$sh"; [ "$sh" = default ] && S=""; [ "$sh" = astream_default ] && S="astream default"
	rm -rf /tmp/pp_$sh && mkdir -p /tmp/pp_$sh
	PLUGIN_BENCH_SHAPE="$S" PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_CLEAN_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$sh -o p -- ./plugin_bench ../../mediastreamer2_amd/libmsmi355xfilters.so 32768 16 200 40 > ../../$OUT/$sh.json 2> ../../$OUT/$sh.err
	f=$(find /tmp/pp_$sh -name "*kernel_stats.csv" | head -1)
	[ -n "$f" ] && cp "$f" ../../$OUT/${sh}_kernel_stats.csv && echo "== $sh" && head -14 ../../$OUT/${sh}_kernel_stats.csv | cut -c1-170
done
