#!/bin/bash
# round 4, GPU pass C: plugin path with the launches leaving at the end of the graph walk; bench line with paced value + plugin_path
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_plugin_fused.py tests/test_gpu_plugin.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -8 | tee gpurun_out/r04c_pytest.log
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
for cfg in "16384 8" "32768 16" "32768 12" "49152 16" "65536 16"; do
  set -- $cfg
  echo "== fused $1 legs / $2 tickers"; timeout 600 $PB $PL $1 $2 300 40 2>/dev/null | tail -1 | tee -a gpurun_out/r04c_plugin_bench.jsonl
done
echo "== no early launch 32768 / 16"; MSMI355X_NO_EARLY_LAUNCH=1 timeout 600 $PB $PL 32768 16 300 40 2>/dev/null | tail -1 | tee -a gpurun_out/r04c_plugin_bench_noearly.jsonl
echo "== bench"
timeout 1500 python bench.py 2>gpurun_out/r04c_bench.err | tee gpurun_out/r04c_bench.json | cut -c1-600
grep -v "sweep" gpurun_out/r04c_bench.err | tail -5
