#!/bin/bash
# round 4, GPU pass K: the slow enqueues call by call (MSMI355X_TRACE_SLOW_MS); SQ counters of volmix_kernel at the headline's scale
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04k_plugin_trace.txt; : > $O
for rep in 1 2; do
  echo "== rep $rep" | tee -a $O
  MSMI355X_TRACE_SLOW_MS=6 timeout 600 $PB $PL 32768 16 1000 40 2>&1 >/tmp/pb.json | grep -v "^ms2shim" | head -80 | tee -a $O
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms')})" | tee -a $O
done
echo "== HSA_ENABLE_SDMA=0" | tee -a $O
HSA_ENABLE_SDMA=0 MSMI355X_TRACE_SLOW_MS=6 timeout 600 $PB $PL 32768 16 1000 40 2>&1 >/tmp/pb.json | grep -v "^ms2shim" | head -40 | tee -a $O
python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms')})" | tee -a $O
echo "== volmix counters"
ST=/tmp/msmi355x_converged.npy
python3 scripts/headline_probe.py 122880 --state $ST > /dev/null 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pmc_k$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_k$i -o pmc -- python3 scripts/headline_probe.py 122880 --state $ST --ticks 8 --settle 8 > /dev/null 2>/tmp/pmc_k$i.err || tail -3 /tmp/pmc_k$i.err
done
python3 - <<'PY' | tee gpurun_out/r04k_volmix_counters.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_k*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "volmix" if "volmix" in k else ("aec_tick" if "aec_tick_kernel" in k else None)
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in agg.items():
    print(name, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
