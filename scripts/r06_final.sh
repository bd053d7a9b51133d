#!/bin/bash
# Runs on the GPU box (via gpurun): what the driver runs at round end -- pytest -m gpu, smoke(), bench.py -- plus the rocprofv3
# kernel stats of the bench command itself.  Results under gpurun_out/r06_final/.
set -u
OUT=gpurun_out/${R06_OUT:-r06_final}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
nproc > $OUT/env.txt; uptime >> $OUT/env.txt; lscpu | grep "Model name" >> $OUT/env.txt
echo "== pytest -m gpu =="
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
echo "== smoke =="
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench =="
timeout 1500 python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.stderr.txt; wc -c $OUT/bench.json; cut -c1-1500 $OUT/bench.json
S=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['streams_per_gpu'])" 2>/dev/null || echo 122880)
echo "== rocprofv3 --kernel-trace --stats of the bench command (headline only, $S legs) =="
rm -rf $OUT/prof && mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python3 bench.py --streams $S --no-cpu-baseline --no-extras --no-plugin-path --no-video-host --no-session --detail $OUT/bench_prof_detail.json > $OUT/bench_prof.json 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv && head -8 $OUT/bench_kernel_stats.csv | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" -delete
rm -rf $OUT/prof
echo "== rocprofv3 --kernel-trace --stats over the extras (every other kernel at the bench sizes) =="
rm -rf $OUT/prof && mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o extras -- python3 bench.py --streams 4096 --steps 16 --warmup 4 --worst-ticks 64 --paced-ticks 0 --zero-ticks 0 --no-cpu-baseline --no-plugin-path --no-video-host --no-session --detail $OUT/bench_extras_detail.json > $OUT/bench_extras.json 2> $OUT/prof_extras.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/all_kernel_stats.csv && head -30 $OUT/all_kernel_stats.csv | cut -c1-160
find $OUT/prof -name "*kernel_trace.csv" -delete
rm -rf $OUT/prof
