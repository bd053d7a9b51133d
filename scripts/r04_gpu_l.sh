#!/bin/bash
# round 4, GPU pass L: volmix parity + rate after the rework
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_mixer.py tests/test_gpu_pipeline.py tests/test_gpu_plugin_fused.py tests/test_gpu_volume.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -5 | tee gpurun_out/r04l_pytest.log
ST=/tmp/msmi355x_converged.npy
python3 scripts/headline_probe.py 122880 --state $ST > /dev/null 2>&1
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o l -- python3 $GRAFT_REPO_ROOT/scripts/headline_probe.py 122880 --state $ST --ticks 32 > /dev/null 2>&1
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -6 | tee $GRAFT_REPO_ROOT/gpurun_out/r04l_kernel_stats.txt
