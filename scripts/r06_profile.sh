#!/bin/bash
# Runs on the GPU box (via gpurun): the headline's chain at a FIXED leg count, in steady state.
#   bash scripts/r06_profile.sh 122880
# 1. rocprofv3 --kernel-trace --stats of scripts/headline_probe.py  -> gpurun_out/r06/headline_fixed_kernel_stats.csv
#    (every aec_tick_kernel<256> launch in it runs at that leg count: frac = 202 240 B x frames / AverageNs / 8 TB/s)
# 2. FETCH_SIZE and WRITE_SIZE in their own passes (no trace domains)  -> gpurun_out/r06/pmc_at_<N>.json
set -u
N=${1:-122880}
OUT=gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ST=/tmp/msmi355x_converged.npy
python3 scripts/headline_probe.py $N --state $ST > $OUT/converge.json 2> $OUT/converge.err || { tail -5 $OUT/converge.err; exit 1; }
cat $OUT/converge.json
rm -rf $OUT/prof && mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o fixed -- python3 scripts/headline_probe.py $N --state $ST --ticks 64 > $OUT/probe_trace.json 2> $OUT/prof.err
cat $OUT/probe_trace.json
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/headline_fixed_kernel_stats.csv && head -8 $OUT/headline_fixed_kernel_stats.csv | cut -c1-220
find $OUT/prof -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf $OUT/pmc_$c && mkdir -p $OUT/pmc_$c
	timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 scripts/headline_probe.py $N --state $ST --ticks 16 --settle 16 > $OUT/pmc_$c/stdout.log 2> $OUT/pmc_$c/stderr.log
done
python3 - $OUT $N <<'PY'
import csv, glob, json, os, re, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if m and "anonymous" in r["Kernel_Name"]:  # this library's kernels
                    agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[c + "_KB"] = round(sum(v) / len(v), 1)
        res[k]["launches"] = len(v)
for k, e in res.items():
    if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE_KB"] * 1024 + e["WRITE_SIZE_KB"] * 1024)
json.dump({"streams": n, "kernels": res, "correction": "read = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B), write = WRITE_SIZE"},
          open(os.path.join(out, "pmc_at_%d.json" % n), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
find $OUT -name "*.csv" -size +8M -delete
# the same counters at BASELINE configs[2]'s 4096 legs (replaces the stale top-level entry of profiles/pmc_summary.json)
if [ "${2:-}" != "" ]; then
  M=$2
  for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf $OUT/pmc_${M}_$c && mkdir -p $OUT/pmc_${M}_$c
	timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${M}_$c -o pmc -- python3 scripts/headline_probe.py $M --state $ST --ticks 16 --settle 16 > $OUT/pmc_${M}_$c/stdout.log 2> $OUT/pmc_${M}_$c/stderr.log
  done
  python3 - $OUT $M <<'PY'
import csv, glob, json, os, re, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "pmc_%d_%s" % (n, c), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if m and "anonymous" in r["Kernel_Name"]:
                    agg[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[c + "_KB"] = round(sum(v) / len(v), 1)
        res[k]["launches"] = len(v)
for k, e in res.items():
    if "FETCH_SIZE_KB" in e and "WRITE_SIZE_KB" in e:
        e["hbm_bytes_per_launch"] = int(2 * e["FETCH_SIZE_KB"] * 1024 + e["WRITE_SIZE_KB"] * 1024)
json.dump({"streams": n, "kernels": res, "correction": "read = 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B), write = WRITE_SIZE"},
          open(os.path.join(out, "pmc_at_%d.json" % n), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
  find $OUT -name "*.csv" -size +8M -delete
fi
