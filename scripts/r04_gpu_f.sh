#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -8 | tee gpurun_out/r04f_pytest.log
echo "== paced probe"
timeout 600 python3 scripts/paced_probe.py 122880 1000 2>&1 | grep -v amdgpu.ids | tail -10 | tee gpurun_out/r04f_paced_probe.txt
