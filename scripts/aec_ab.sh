#!/bin/bash
# A/B on the GPU box: rebuild aec.o with each set of -D flags given as arguments (quote each set), probe the chain
set -u
mkdir -p gpurun_out
: > gpurun_out/aec_ab.log
for defs in "$@"; do
  rm -f mediastreamer2_amd/csrc/aec.o
  make -C mediastreamer2_amd/csrc -j8 DEFS="$defs" > gpurun_out/aec_ab_build.log 2>&1 || tail -5 gpurun_out/aec_ab_build.log
  echo "== $defs" | tee -a gpurun_out/aec_ab.log
  NO_OVERLAP=1 bash scripts/prof_pipe.sh ${N:-65536} | sed -n 2,3p | cut -d, -f1-4,6,7 | cut -c30-200 | tee -a gpurun_out/aec_ab.log
  python3 scripts/pipe_probe.py ${N:-65536} 2>/dev/null | cut -c1-130 | tee -a gpurun_out/aec_ab.log
done
