#!/bin/bash
# Dev tool, on the GPU box: where the canceller's VALU instructions go.  Rebuilds aec.o with a phase switched off
# (results are then wrong: instruction counts only) and reads SQ_INSTS_VALU / SQ_WAVE_CYCLES for 9 launches of 4096 legs.
set -u
mkdir -p gpurun_out
: > gpurun_out/aec_phase.log
for defs in "" "-DAEC_PROF_NO_FFT" "-DAEC_PROF_NO_STREAM_MATH" "-DAEC_PROF_NO_FFT -DAEC_PROF_NO_STREAM_MATH"; do
  for post in 1 0; do
    rm -f mediastreamer2_amd/csrc/aec.o
    make -C mediastreamer2_amd/csrc -j8 DEFS="$defs" > gpurun_out/aec_phase_build.log 2>&1 || tail -5 gpurun_out/aec_phase_build.log
    echo "== defs='$defs' postfilter=$post" | tee -a gpurun_out/aec_phase.log
    AEC_PROBE_POST=$post bash scripts/pmc_run.sh pmc_aec_phase "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" python3 scripts/pmc_probe.py aec | grep -v "^void" | tee -a gpurun_out/aec_phase.log
  done
done
rm -f mediastreamer2_amd/csrc/aec.o
make -C mediastreamer2_amd/csrc -j8 > /dev/null 2>&1
