#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, a bench line, rocprof kernel stats.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
echo "== env ==" > gpurun_out/env.log
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
echo "== pytest -m gpu =="
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== bench =="
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
S=$(python3 -c "import json;print(json.load(open('gpurun_out/bench.json'))['config']['streams_per_gpu'])" 2>/dev/null || echo 65536)
echo "== rocprof kernel stats (headline at $S legs) =="
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
# pass 1: the headline workload alone (the chain + the canceller leg of the roofline), at the capacity the bench found
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o headline -- python3 bench.py --streams $S --no-cpu-baseline --no-extras > gpurun_out/bench_prof_headline.json 2> gpurun_out/prof.err
# pass 2: every kernel of the path at the bench sizes (other_kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o all -- python3 bench.py --streams 4096 --no-cpu-baseline --no-session > gpurun_out/bench_prof_all.json 2>> gpurun_out/prof.err
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "$f"; head -14 "$f" | cut -c1-200; done
# keep the big traces out of the merge budget
find gpurun_out/prof -name "*kernel_trace.csv" -delete
