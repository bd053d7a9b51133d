#!/bin/bash
# A/B on the GPU box: the 256-sample canceller built for 3 (default) and 2 waves per SIMD
set -u
mkdir -p gpurun_out
for occ in 2 3; do
  rm -f mediastreamer2_amd/csrc/aec.o
  make -C mediastreamer2_amd/csrc -j8 DEFS=-DAEC_OCC256=$occ > /dev/null 2>&1
  echo "occ $occ" | tee -a gpurun_out/aec_occ_ab.log
  python3 scripts/aec_rate_probe.py 65536 2>&1 | tail -1 | tee -a gpurun_out/aec_occ_ab.log
  python3 scripts/pipe_probe.py 65536 2>&1 | tail -1 | tee -a gpurun_out/aec_occ_ab.log
done
