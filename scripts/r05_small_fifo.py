"""8 / 16 kHz legs through the canceller's FIFO entry (round 5): whole 10 ms ticks of 65 536 legs, group form (aec_fifos_group ->
aec_group_kernel) against the tick form one leg per wavefront (MSMI355X_AEC_GROUP=0).  One line per run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mediastreamer2_amd as ms  # noqa: E402

ctx = ms.Context(0)
for rate, F in ((8000, 64), (16000, 128)):
    lg = bench.make_aec_small_fifo_leg(ms, torch, ctx, rate, F, int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
    g = lg.run(96, 8, use_graph=True)
    ctx.sync()
    reps = [lg.timed(96, g) for _ in range(5)]
    r = bench.roofline(min(reps), 96, lg.alg_bytes)
    print(json.dumps({"group_form": os.environ.get("MSMI355X_AEC_GROUP", "1"), "rate": rate, "F": F, "legs": lg.units, "tick_us": r["avg_launch_us"],
                      "algorithmic_GBps": r["achieved"], "frac": r["frac"], "replays_us": [round(x * 1e3 / 96, 1) for x in reps]}))
    del lg, g
    torch.cuda.empty_cache()
