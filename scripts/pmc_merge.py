"""Fold the counter passes collected AT a fixed leg count (scripts/r03_profile.sh -> gpurun_out/r03/pmc_at_<N>.json) into
profiles/pmc_summary.json: kernel -> "at_streams" -> "<N>" -> {FETCH_SIZE_KB, WRITE_SIZE_KB, hbm_bytes_per_launch, launches}.
bench.py reads the canceller's entry for roofline.traffic (pmc_traffic_at).   python3 scripts/pmc_merge.py gpurun_out/r03/pmc_at_*.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "pmc_summary.json")
summary = json.load(open(path)) if os.path.exists(path) else {}
for f in sys.argv[1:]:
    d = json.load(open(f))
    n = str(d["streams"])
    for k, e in d["kernels"].items():
        if "hbm_bytes_per_launch" not in e:
            continue
        entry = dict(e, state="steady state (converged cancellers, SURVEY 8(d) echo scene), product stagger", correction=d["correction"])
        summary.setdefault(k, {}).setdefault("at_streams", {})[n] = entry
json.dump(summary, open(path, "w"), indent=1, sort_keys=True)
print("merged", sys.argv[1:])
