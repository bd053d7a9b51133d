"""Dev tool: all-kernels-per-tick probe at several stream counts."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
for n in [int(a) for a in sys.argv[1:]] or (49152,):
    print(json.dumps(bench.pipeline_probe(ms, torch, ctx, n)), flush=True)
