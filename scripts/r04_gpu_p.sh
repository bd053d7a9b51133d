#!/bin/bash
# round 4, GPU pass P: the whole GPU suite and the whole bench line on the current tree
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -8 | tee gpurun_out/r04p_pytest.log
timeout 1800 python bench.py 2>gpurun_out/r04p_bench.err | tee gpurun_out/r04p_bench.json | cut -c1-300
grep -v "sweep" gpurun_out/r04p_bench.err | tail -5
