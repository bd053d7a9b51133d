#!/bin/bash
# round 4, GPU pass N: the leg bank's tick path without copies (default) / copies as kernels / hipMemcpyAsync; the group form of the small-frame cancellers
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04n_plugin_copies.txt; : > $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "plugin or fused or mixer or volume or pipeline" 2>&1 | grep -v "^ms2shim" | tail -5 | tee gpurun_out/r04n_pytest.log
one() { echo "== $*" | tee -a $O; env "$@" timeout 600 $PB $PL 32768 16 1000 40 2>/dev/null | tail -1 > /tmp/pb.json
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','p99_9_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms','max_backlog_ms','launches_per_tick_and_ticker')})
for s in d['slow_ticks'][:2]: print('   ',s)" | tee -a $O; }
for rep in 1 2; do
  one A=1
  one MSMI355X_ZERO_COPY=0
  one MSMI355X_ZERO_COPY=0 MSMI355X_COPY=hip
  one GPU_MAX_HW_QUEUES=16
done
echo "== AEC group form"
timeout 1500 python -m pytest tests/test_gpu_aec.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -15 | tee gpurun_out/r04n_pytest_aec.log
