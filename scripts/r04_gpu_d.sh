#!/bin/bash
# round 4, GPU pass D: full GPU suite on the specialised tick kernels; A/B of the row-wise chains; plugin path; paced probe; bench
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -15 | tee gpurun_out/r04d_pytest.log
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
for cfg in "16384 8" "32768 16" "32768 12" "49152 16" "65536 16"; do
  set -- $cfg
  echo "== fused $1 legs / $2 tickers"; timeout 600 $PB $PL $1 $2 300 40 2>/dev/null | tail -1 | tee -a gpurun_out/r04d_plugin_bench.jsonl
done
echo "== no early launch 32768 / 16"; MSMI355X_NO_EARLY_LAUNCH=1 timeout 600 $PB $PL 32768 16 300 40 2>/dev/null | tail -1 | tee -a gpurun_out/r04d_plugin_bench_noearly.jsonl
echo "== A/B chains"
cp mediastreamer2_amd/libmsmi355x.so /tmp/lib_default.so
for rep in 1 2; do
  for m in 0 11 15; do AB_STEADY=1 bash scripts/ab_build.sh "rows=$m" "-DAEC_CHAIN_ROWS=$m" 122880 2>&1 | tail -1 | tee -a gpurun_out/r04d_ab_chains.txt; done
done
touch mediastreamer2_amd/csrc/aec.hip; make -C mediastreamer2_amd/csrc > /dev/null 2>&1
echo "== paced probe"
timeout 600 python3 scripts/paced_probe.py 122880 1000 2>&1 | tail -9 | tee gpurun_out/r04d_paced_probe.txt
echo "== bench"
timeout 1500 python bench.py 2>gpurun_out/r04d_bench.err | tee gpurun_out/r04d_bench.json | cut -c1-300
grep -v "sweep" gpurun_out/r04d_bench.err | tail -5
