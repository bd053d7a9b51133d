#!/bin/bash
# where the first ticks after an attach go: the ticker threads' stacks when a warm-up step has run for 40 ms, and the leg bank's own trace
# usage: scripts/r06_stall_probe.sh [shape] [legs]
make -C tests/host plugin_bench >/dev/null 2>&1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
PLUGIN_BENCH_SHAPE="${1:-astream}" PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_STACKS=${3:-40} PLUGIN_BENCH_STACKS_WARMUP=1 MSMI355X_TRACE_SLOW_MS=20 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so ${2:-32768} $T 60 20 > gpurun_out/r06_stall.json 2> gpurun_out/r06_stall_stderr.txt
grep -v "ms2shim-warning" gpurun_out/r06_stall_stderr.txt | head -220 > gpurun_out/r06_stall.txt
cut -c1-200 gpurun_out/r06_stall.txt
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_stall.json').read().strip().splitlines()[-1]); print(d['from_attach'])"
