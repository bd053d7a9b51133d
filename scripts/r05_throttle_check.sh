#!/bin/bash
# is the plugin path's noise the cgroup's CPU quota?  cpu.stat (nr_throttled, throttled_usec) around plugin_bench runs, by ticker count
set -u
make -C tests/host -s plugin_bench >/dev/null 2>&1
P=mediastreamer2_amd/libmsmi355xfilters.so
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ' || cat /sys/fs/cgroup/cpu/cpu.stat | tr '\n' ' '; echo; }
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  loadavg: $(cat /proc/loadavg)"
for T in 16 12 8; do
  echo "== $T tickers, 32768 legs"; stat
  PLUGIN_BENCH_PACED=1 tests/host/plugin_bench $P 32768 $T 600 40 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('p50', d['p50_ms'], 'p99', d['p99_ms'], 'max', d['max_ms'], 'late', d['late'], 'us/leg-tick', d['us_per_leg_tick'], 'cpu_ms/ticker-tick', d['ticker_cpu_ms'], 'nivcsw', d['nivcsw_per_tick_and_ticker'], 'slow', d['slow_ticks'][:2])"
  stat
done
echo "loadavg: $(cat /proc/loadavg)"; top -b -n 1 | head -15
