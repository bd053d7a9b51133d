"""Dev probe: the chain's tick at N legs with the legs' re-framing phases aligned vs spread (bench.py's two rigs).
  python scripts/stagger_probe.py 90112 94208"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

ctx = ms.Context(0)
for n in [int(v) for v in sys.argv[1:]] or [65536]:
    for st in (False, True):
        p = bench.chain_capacity_point(ms, torch, ctx, n, stagger=st)
        p["stagger"] = st
        print(json.dumps(p), flush=True)
