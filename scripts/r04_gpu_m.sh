#!/bin/bash
# round 4, GPU pass M: the plugin's copies as kernels (default) against hipMemcpyAsync (MSMI355X_COPY=hip); plugin GPU tests; volmix
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04m_plugin_copies.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "plugin or fused or mixer or volume or pipeline" 2>&1 | grep -v "^ms2shim" | tail -5 | tee gpurun_out/r04m_pytest.log
for rep in 1 2 3; do
 for mode in kernel hip; do
  echo "== rep $rep copies: $mode" | tee -a $O
  MSMI355X_COPY=$mode timeout 600 $PB $PL 32768 16 1000 40 2>/dev/null | tail -1 > /tmp/pb.json
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','p99_9_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms','max_backlog_ms')})
for s in d['slow_ticks'][:3]: print('   ',s)" | tee -a $O
 done
done
bash scripts/r04_gpu_l.sh
