#!/bin/bash
# round 4, GPU pass B: the fused call-leg chain behind the plugin -- parity tests, then the plugin-path rate
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
nproc > gpurun_out/r04b_env.txt; lscpu | grep "Model name" >> gpurun_out/r04b_env.txt
timeout 1800 python -m pytest tests/test_gpu_plugin_fused.py tests/test_gpu_plugin.py tests/test_gpu_plugin_codec.py tests/test_aec_tester_scenarios.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -40 | tee gpurun_out/r04b_pytest.log
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
for cfg in "4096 4" "16384 8" "32768 8" "32768 16" "65536 16"; do
  set -- $cfg
  echo "== fused $1 legs / $2 tickers"; timeout 600 $PB $PL $1 $2 300 40 2>/dev/null | tail -1 | tee -a gpurun_out/r04b_plugin_bench.jsonl
done
echo "== one by one 4096 / 4"; MSMI355X_NO_FUSE=1 timeout 600 $PB $PL 4096 4 100 20 2>/dev/null | tail -1 | tee -a gpurun_out/r04b_plugin_bench_nofuse.jsonl
echo "== one by one 16384 / 8"; MSMI355X_NO_FUSE=1 timeout 600 $PB $PL 16384 8 60 20 2>/dev/null | tail -1 | tee -a gpurun_out/r04b_plugin_bench_nofuse.jsonl
