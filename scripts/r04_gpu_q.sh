#!/bin/bash
# round 4: the small-frame cancellers -- parity tests, then launch time with several legs per wavefront and with one
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_gpu_aec.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -4 | tee gpurun_out/r04q_pytest.log
for rep in 1 2; do
  for gform in 1 0; do echo "== MSMI355X_AEC_GROUP=$gform"; MSMI355X_AEC_GROUP=$gform timeout 600 python3 scripts/aec_rate_probe.py 65536 2>/dev/null | head -2; done
done | tee gpurun_out/r04q_group_rate.txt
