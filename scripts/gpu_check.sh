#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, a bench line, rocprof kernel stats.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== env ==" > gpurun_out/env.log
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== bench =="
timeout 600 python bench.py --steps 400 --warmup 20 2>gpurun_out/bench.err | tee gpurun_out/bench.json
echo "== rocprof kernel stats =="
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
# pass 1: the headline workload alone, so the kernel's average in the summary is the bench's launch
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o headline -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_headline.json 2> gpurun_out/prof.err
# pass 2: every kernel of the path (the resampler row then mixes the 4096- and 65536-stream launches)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o all -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 > gpurun_out/bench_prof_all.json 2>> gpurun_out/prof.err
# pass 3: the all-kernels-per-tick probe at 65536 streams (north_star check), on its own
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o pipeline -- python3 scripts/pipe_probe.py 65536 > gpurun_out/pipe_prof.json 2>> gpurun_out/prof.err
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "$f"; head -12 "$f" | cut -c1-200; done
# keep the big traces out of the merge budget
find gpurun_out/prof -name "*kernel_trace.csv" -delete
