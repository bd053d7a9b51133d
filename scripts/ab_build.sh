#!/bin/bash
# Dev tool (GPU box): rebuild libmsmi355x.so's canceller with extra -D flags and time the chain.
#   bash scripts/ab_build.sh "<label>" "<defs>" N
set -u
label=$1; defs=$2; n=${3:-118784}
touch mediastreamer2_amd/csrc/aec.hip
make -C mediastreamer2_amd/csrc DEFS="$defs" > /dev/null 2>&1 || { echo "build failed: $label"; exit 1; }
for i in 1; do
python3 - "$label" "$n" <<'PY'
import sys, json
sys.path.insert(0, ".")
import torch, bench, mediastreamer2_amd as ms
ctx = ms.Context(0)
import os
conv = bench.Converged(ms, torch, ctx) if os.environ.get("AB_STEADY") else None
p = bench.chain_capacity_point(ms, torch, ctx, int(sys.argv[2]), worst_ticks=96, converged=conv)
print(sys.argv[1], json.dumps({k: p[k] for k in ("streams", "tick_ms_avg", "tick_ms_worst", "tick_ms_single_median")}), flush=True)
PY
done
