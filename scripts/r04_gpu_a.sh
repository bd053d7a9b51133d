#!/bin/bash
# round 4, GPU pass A: full GPU test suite, then a bench line (the list hand-over is now two launches)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 | tee gpurun_out/r04a_pytest_gpu.log
timeout 900 python bench.py --no-session 2>gpurun_out/r04a_bench.err | tee gpurun_out/r04a_bench.json
tail -30 gpurun_out/r04a_bench.err
