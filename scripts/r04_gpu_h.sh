#!/bin/bash
# round 4, GPU pass H: the full bench line on the tree of the committed profiles
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python bench.py 2>gpurun_out/r04h_bench.err | tee gpurun_out/r04h_bench.json | cut -c1-400
grep -v "sweep" gpurun_out/r04h_bench.err | tail -5
