"""Dev tool for PMC passes: a few plain (non-graph) launches of one kernel.  python3 scripts/pmc_probe.py resample 262144"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mk = {"resample": lambda: bench.make_resample_leg(ms, torch, ctx, n or 4096),
      "mixer": lambda: bench.make_mixer_leg(ms, torch, ctx, nconf=n or 128),
      "volume": lambda: bench.make_volume_leg(ms, torch, ctx, nstreams=n or 4096),
      "equalizer": lambda: bench.make_equalizer_leg(ms, torch, ctx),
      "aec": lambda: bench.make_aec_leg(ms, torch, ctx, n or 4096), "scaler": lambda: bench.make_scaler_leg(ms, torch, ctx), "pixconv": lambda: bench.make_pixconv_leg(ms, torch, ctx),
      "g711dec": lambda: bench.make_g711_leg(ms, torch, ctx), "g711enc": lambda: bench.make_g711_leg(ms, torch, ctx, encode=True),
      "plc": lambda: bench.make_plc_leg(ms, torch, ctx)}
lg = mk[which]()
for i in range(9 if which == "aec" else 6):  # the canceller's ring is one 8-tick cycle (15 frames per leg)
    lg.launch(i % lg.ring)
ctx.sync()
