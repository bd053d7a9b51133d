#!/bin/bash
# kernel-trace stats of the chained tick at N legs (default 65536), canceller / post-filter not overlapped (clean durations)
set -u
N=${1:-65536}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_pipe && mkdir -p gpurun_out/prof_pipe
MSMI355X_AEC_NO_OVERLAP=${NO_OVERLAP:-1} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pipe -o pipe -- python3 scripts/pipe_probe.py $N > gpurun_out/prof_pipe/out.json 2> gpurun_out/prof_pipe/err.log
f=$(find gpurun_out/prof_pipe -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-180
find gpurun_out/prof_pipe -name "*kernel_trace.csv" -delete
