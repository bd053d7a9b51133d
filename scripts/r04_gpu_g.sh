#!/bin/bash
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -8 | tee gpurun_out/r04g_pytest.log
echo "== profile"
bash scripts/r04_profile.sh 122880 4096 2>&1 | tail -40
