#!/bin/bash
# usage: pmc_run.sh <outdir> "<counters>" <program args...>   (counters in their own pass; no trace domains)
set -u
out=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf "gpurun_out/$out"; mkdir -p "gpurun_out/$out"
rocprofv3 --pmc $ctrs --output-format csv -d "gpurun_out/$out" -o pmc -- "$@" > "gpurun_out/$out/stdout.log" 2> "gpurun_out/$out/stderr.log"
f=$(find "gpurun_out/$out" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "aec" in k or "resample" in k or "scaler" in k or "mixer" in k or "volume" in k or "equalizer" in k or "pixconv" in k or "fifo" in k or "g711" in k or "plc" in k:
        print(k)
        for c, v in d.items():
            print("   %-28s n=%4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
find "gpurun_out/$out" -name "*.csv" -size +8M -delete
