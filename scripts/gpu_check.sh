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
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
find gpurun_out/prof -name "*stats*" | head; 
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
# keep the big traces out of the merge budget
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
