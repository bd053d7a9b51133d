#!/bin/bash
# where the ticker threads are during the FIRST ticks after an attach (the sampler of r06_walk_profile.sh over the warm-up ticks only)
# usage: scripts/r06_first_ticks_profile.sh [legs=16384] [warm-up ticks=4] [shape]
make -C tests/host plugin_bench >/dev/null 2>&1
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
LEGS=${1:-16384}; W=${2:-4}; SHAPE="${3:-}"
PLUGIN_BENCH_SHAPE="$SHAPE" PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SAMPLE=4000 PLUGIN_BENCH_SAMPLE_WARMUP=1 timeout 300 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $LEGS $T 60 $W > gpurun_out/r06_first_ticks.json 2> gpurun_out/r06_first_ticks.stderr
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_first_ticks.json').read().strip().splitlines()[-1]); print(d['from_attach'])"
{ echo "== the first $W ticks after the attach: shape '$SHAPE' legs $LEGS tickers $T"; python3 scripts/walk_profile.py gpurun_out/r06_first_ticks.stderr 45; } > gpurun_out/r06_first_ticks.txt
rm -f gpurun_out/r06_first_ticks.stderr
head -75 gpurun_out/r06_first_ticks.txt
