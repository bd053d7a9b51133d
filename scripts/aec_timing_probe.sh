#!/bin/bash
# on the GPU box: rebuild with the phase stamps, run the probe, rebuild the product library
set -u
rm -f mediastreamer2_amd/csrc/aec.o
make -C mediastreamer2_amd/csrc -j8 DEFS="-DAEC_PROF_TIMING ${EXTRA_DEFS:-}" > gpurun_out/aec_timing_build.log 2>&1 || tail -5 gpurun_out/aec_timing_build.log
python3 scripts/aec_timing_probe.py ${N:-65536} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/aec_timing.log
rm -f mediastreamer2_amd/csrc/aec.o
make -C mediastreamer2_amd/csrc -j8 > /dev/null 2>&1
