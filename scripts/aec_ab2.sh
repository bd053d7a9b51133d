#!/bin/bash
# A/B on the GPU box: rebuild aec.o with each set of -D flags given as arguments (quote each set); two-frame and one-frame
# launch time of the canceller at N legs (scripts/aec_mix_probe.py), variants interleaved twice against clock drift
set -u
mkdir -p gpurun_out
: > gpurun_out/aec_ab2.log
for rep in 1 2; do
for defs in "$@"; do
  rm -f mediastreamer2_amd/csrc/aec.o
  make -C mediastreamer2_amd/csrc -j8 DEFS="$defs" > gpurun_out/aec_ab_build.log 2>&1 || tail -5 gpurun_out/aec_ab_build.log
  echo "== [$rep] $defs" | tee -a gpurun_out/aec_ab2.log
  python3 scripts/aec_mix_probe.py ${N:-65536} 2>/dev/null | head -2 | tee -a gpurun_out/aec_ab2.log
done
done
rm -f mediastreamer2_amd/csrc/aec.o
make -C mediastreamer2_amd/csrc -j8 > /dev/null 2>&1
