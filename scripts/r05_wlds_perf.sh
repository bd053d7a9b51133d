#!/bin/bash
# W-in-LDS experiment (round 5): the headline tick at a FIXED 122 880 legs -- product form, the experiment, and the product form with
# LDS it never touches (what the footprint alone costs: 2 and 3 wavefronts per CU) -- bench.py's roofline object for each, then
# rocprofv3 kernel stats of the first two
set -u
mkdir -p gpurun_out/r05w
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="python3 bench.py --streams 122880 --no-extras --no-cpu-baseline --no-plugin-path --no-video-host --no-session --paced-ticks 0 --worst-ticks 600 --zero-ticks 0"
run() { # name, env...
  n=$1; shift
  env "$@" $B --detail gpurun_out/r05w/$n.detail.json > gpurun_out/r05w/$n.json 2> gpurun_out/r05w/$n.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r05w/$n.json')); r=d['roofline']; print('$n', 'avg_launch_us', r['avg_launch_us'], 'frac', r['frac'], 'ms_per_step', d['ms_per_step'], 'p50', d['config']['consecutive']['p50_ms'], 'max', d['config']['consecutive']['max_ms'])"
}
run product A=1
run w_in_lds MSMI355X_AEC_W_IN_LDS=1
run pad_48k MSMI355X_AEC_LDS_PAD=49152
run pad_24k MSMI355X_AEC_LDS_PAD=24576
for v in product w_in_lds; do
  E="A=1"; [ $v = w_in_lds ] && E="MSMI355X_AEC_W_IN_LDS=1"
  rm -rf gpurun_out/r05w/prof_$v
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05w/prof_$v -o $v -- $B --worst-ticks 200 > /dev/null 2> gpurun_out/r05w/prof_$v.err
  f=$(find gpurun_out/r05w/prof_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r05w/${v}_kernel_stats.csv && head -4 $f | cut -c1-220
  find gpurun_out/r05w/prof_$v -name "*kernel_trace.csv" -delete
done
