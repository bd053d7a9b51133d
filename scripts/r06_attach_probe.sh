#!/bin/bash
# Round 6: the first ticks after an attach (from_attach) and the steady state, for several shapes, paced, 16 tickers.
# usage: scripts/r06_attach_probe.sh <tag> [legs=32768]
tag=${1:-x}; legs=${2:-32768}
out=gpurun_out/r06_attach_${tag}.txt
make -C tests/host plugin_bench >/dev/null 2>&1
uptime > $out
T=$(python3 -c "import os;print(min(16,len(os.sched_getaffinity(0))))")
for shape in "astream" "" "nomixer noagc" "server dec"; do
  echo "== shape '$shape' legs $legs" >> $out
  PLUGIN_BENCH_SHAPE="$shape" PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so $legs $T 250 40 >> $out 2>&1
done
echo "== shape '' legs 16384" >> $out
PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 16384 $T 250 40 >> $out 2>&1
uptime >> $out
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l)
        print({k:d[k] for k in ("legs","p50_ms","p99_ms","max_ms","late","us_per_leg_tick","ticker_flush_ms","ticker_graph_walk_ms","fused_legs","late_events","build_ms")}, d["from_attach"])
PY
echo "== churn: 20 re-plumbings a second per ticker, 2048 legs per ticker" | tee -a $out
PLUGIN_BENCH_CHURN=20 PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 $T 400 40 2>/dev/null | tee -a $out | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','fused_legs','late_events','churn')})"
echo "== churn, server dec shape" | tee -a $out
PLUGIN_BENCH_SHAPE="server dec" PLUGIN_BENCH_CHURN=20 PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 $T 400 40 2>/dev/null | tee -a $out | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','fused_legs','late_events','churn')})"
echo "== churn on the ticker's own thread (round 6's first form), for comparison" | tee -a $out
PLUGIN_BENCH_CHURN_ON_TICKER=1 PLUGIN_BENCH_CHURN=20 PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 $T 400 40 2>/dev/null | tee -a $out | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','fused_legs','late_events','churn')})"
echo "== churn, astream default shape (one stream's graph at a time)" | tee -a $out
PLUGIN_BENCH_SHAPE="astream default" PLUGIN_BENCH_CHURN=20 PLUGIN_BENCH_PACED=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 16384 $T 400 40 2>/dev/null | tee -a $out | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('legs','p50_ms','p99_ms','max_ms','late','us_per_leg_tick','fused_legs','late_events','churn')})"
