#!/bin/bash
# Dev probe (GPU box): the device's accumulated throttle residencies (amd-smi metric --throttle: power limit PPT, the
# thermal ones, PROCHOT) before and after two 3000-tick series of the chain, and once a second while they run.
#   bash scripts/throttle_probe.sh 131072
N=${1:-131072}
acc() { amd-smi metric -g 0 --throttle 2>/dev/null | grep -E "ACCUMULATION_COUNTER|PROCHOT_ACCUMULATED|PPT_ACCUMULATED|SOCKET_THERMAL_ACCUMULATED|VR_THERMAL_ACCUMULATED|HBM_THERMAL_ACCUMULATED" | tr -s ' ' | tr '\n' ' '; echo; }
echo "before: $(acc)"
( while true; do echo "$(date +%s.%N | cut -c1-14) $(acc) $(amd-smi metric -g 0 --power 2>/dev/null | grep SOCKET_POWER | tr -s ' ') $(amd-smi metric -g 0 --clock 2>/dev/null | grep -A1 "GFX_[0-7]:" | grep " CLK:" | tr -s ' ' | tr '\n' ' ')"; sleep 0.5; done ) > /tmp/throttle_poll.txt &
POLL=$!
python3 - $N <<'PY'
import sys, json, time
sys.path.insert(0, ".")
import numpy as np, torch, bench, mediastreamer2_amd as ms
n = int(sys.argv[1])
ctx = ms.Context(0)
conv = bench.Converged(ms, torch, ctx)
head = bench.Headline(ms, torch, ctx, n, 1, 0, None, 0)
head.prepare(16, conv)
head.tick_series(16)
for rep in range(3):
    t0 = time.time()
    if rep < 2:
        v = head.tick_series(3000)
        label = "back to back"
    else:  # paced: a tick every 10 ms of wall time, as an MSTicker runs them (the GPU idles for the rest of the interval)
        v = np.empty(3000)
        nxt = time.perf_counter()
        for t in range(3000):
            while time.perf_counter() < nxt:
                pass
            nxt += 0.010
            ctx.timer_start()
            head.g1[t % len(head.g1)].launch()
            v[t] = ctx.timer_stop()
        label = "one tick per 10 ms of wall time"
    top = np.argsort(v)[-4:][::-1]
    print(json.dumps({"rep": label, "t0": round(t0, 2), "wall_s": round(time.time() - t0, 2), **bench.series_stats(v)}), flush=True)
PY
kill $POLL
echo "after: $(acc)"
echo "polls: $(wc -l < /tmp/throttle_poll.txt)"; awk '{ppt=""; pw=""; clk=""; for(i=1;i<=NF;i++) {if($i=="PPT_ACCUMULATED:") ppt=$(i+1); if($i=="SOCKET_POWER:") pw=$(i+1); if($i=="CLK:") clk=clk" "$(i+1)}; print $1, "ppt", ppt, "W", pw, "gfx MHz", clk}' /tmp/throttle_poll.txt | awk 'NR%6==1' | head -40
