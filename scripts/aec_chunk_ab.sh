#!/bin/bash
# A/B on the GPU box: post-filter overlap chunks for the tick-form canceller, chain at 65536 / 98304 legs
set -u
mkdir -p gpurun_out
rm -f gpurun_out/aec_chunk_ab.log
for ch in 2 3 4 6; do
  echo "chunks $ch" | tee -a gpurun_out/aec_chunk_ab.log
  MSMI355X_AEC_CHUNKS=$ch python3 scripts/pipe_probe.py 65536 98304 2>&1 | cut -c1-260 | tee -a gpurun_out/aec_chunk_ab.log
done
echo "no overlap" | tee -a gpurun_out/aec_chunk_ab.log
MSMI355X_AEC_NO_OVERLAP=1 python3 scripts/pipe_probe.py 65536 2>&1 | cut -c1-260 | tee -a gpurun_out/aec_chunk_ab.log
