#!/bin/bash
# round 4, GPU pass J: where the plugin path's rare ~13 ms steps go (per-filter-id profile of the slowest step; the host's NUMA / THP counters around a run)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
PB=tests/host/plugin_bench; PL=mediastreamer2_amd/libmsmi355xfilters.so
O=gpurun_out/r04j_plugin_diag.txt; : > $O
{
echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>&1)"; echo "thp: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>&1)"
ls -d /sys/devices/system/node/node* 2>/dev/null | tr '\n' ' '; echo
grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status
vm() { grep -E "^(numa_hint_faults|numa_pages_migrated|numa_pte_updates|thp_fault_alloc|thp_collapse_alloc|compact_stall|pgmigrate_success|pgfault|nr_tlb_remote_flush|nr_tlb_local_flush_all) " /proc/vmstat | tr '\n' ' '; echo; }
for rep in 1 2 3; do
  echo "== rep $rep: 32768 legs / 16 tickers, profile on"; vm
  MS2SHIM_PROFILE=1 timeout 600 $PB $PL 32768 16 1000 40 2>&1 >/tmp/pb.json | grep "plugin_bench profile"
  vm
  python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms','minflt_per_tick_and_ticker')})
for s in d['slow_ticks']: print('   ',s)"
done
echo "== GPU_MAX_HW_QUEUES=16, profile on"
GPU_MAX_HW_QUEUES=16 MS2SHIM_PROFILE=1 timeout 600 $PB $PL 32768 16 1000 40 2>&1 >/tmp/pb.json | grep "plugin_bench profile"
python3 -c "
import json; d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('p50_ms','p99_ms','max_ms','late','ticker_graph_walk_ms','ticker_flush_ms','minflt_per_tick_and_ticker')})
for s in d['slow_ticks']: print('   ',s)"
} 2>&1 | tee -a $O
echo "== volmix parity + rate"
timeout 900 python -m pytest tests/test_gpu_mixer.py tests/test_gpu_pipeline.py tests/test_gpu_plugin_fused.py -m gpu -q -x 2>&1 | grep -v "^ms2shim" | tail -5 | tee gpurun_out/r04j_pytest.log
timeout 600 python3 scripts/headline_probe.py 122880 --ticks 32 2>/dev/null | tail -2 | tee gpurun_out/r04j_probe.json
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j -o j -- python3 $GRAFT_REPO_ROOT/scripts/headline_probe.py 122880 --ticks 32 > /dev/null 2>&1
f=$(find /tmp/prof_j -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -6 | tee $GRAFT_REPO_ROOT/gpurun_out/r04j_kernel_stats.txt
