#!/bin/bash
# round 4: what the driver runs at the round's end, on the current tree -- the GPU suite, smoke(), the bench line
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -8 | tee gpurun_out/r04p_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v "^ms2shim" | tail -3 | tee gpurun_out/r04p_smoke.log
timeout 1800 python bench.py 2>gpurun_out/r04p_bench.err | tee gpurun_out/r04p_bench.json | cut -c1-300
grep -v "sweep" gpurun_out/r04p_bench.err | tail -5
