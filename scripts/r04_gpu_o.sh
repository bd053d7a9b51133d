#!/bin/bash
# round 4, GPU pass O: launch time of the small-frame cancellers, group form against one leg per wavefront
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for rep in 1 2; do
  for gform in 1 0; do echo "== MSMI355X_AEC_GROUP=$gform"; MSMI355X_AEC_GROUP=$gform timeout 600 python3 scripts/aec_rate_probe.py 65536 2>/dev/null | head -2; done
done | tee gpurun_out/r04o_group_rate.txt
