"""Per-kernel GPU timing outside bench.py (dev tool): python scripts/kbench.py scaler|mixer|volume|equalizer|resample|aec"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mediastreamer2_amd as ms
import bench
ctx = ms.Context(0)
which = sys.argv[1:] or ["resample", "mixer", "volume", "equalizer", "aec", "scaler"]
mk = {"resample": lambda: bench.make_resample_leg(ms, torch, ctx, 4096), "mixer": lambda: bench.make_mixer_leg(ms, torch, ctx),
      "volume": lambda: bench.make_volume_leg(ms, torch, ctx), "equalizer": lambda: bench.make_equalizer_leg(ms, torch, ctx),
      "aec": lambda: bench.make_aec_leg(ms, torch, ctx), "scaler": lambda: bench.make_scaler_leg(ms, torch, ctx), "pixconv": lambda: bench.make_pixconv_leg(ms, torch, ctx),
      "g711_dec": lambda: bench.make_g711_leg(ms, torch, ctx), "g711_enc": lambda: bench.make_g711_leg(ms, torch, ctx, encode=True),
      "ulaw_dec": lambda: bench.make_g711_leg(ms, torch, ctx, law=ms.MI_LAW_PCMU), "ulaw_enc": lambda: bench.make_g711_leg(ms, torch, ctx, law=ms.MI_LAW_PCMU, encode=True),
      "g711_dec_8k": lambda: bench.make_g711_leg(ms, torch, ctx, n=80),
      "plc": lambda: bench.make_plc_leg(ms, torch, ctx), "plc_clean": lambda: bench.make_plc_leg(ms, torch, ctx, loss=0.0),
      "plc_half": lambda: bench.make_plc_leg(ms, torch, ctx, loss=0.5), "plc48": lambda: bench.make_plc_leg(ms, torch, ctx, nstreams=16384, rate=48000),
      "scaler_i420": lambda: bench.make_scaler_leg(ms, torch, ctx, fmt=ms.MI_PIX_I420),
      "pixconv_rgb": lambda: bench.make_pixconv_leg(ms, torch, ctx, fmt=ms.MI_PIX_BGR24)}
for w in which:
    lg = mk[w]()
    K = 100
    g = lg.run(K, 3)
    ctx.sync()
    best = min(lg.timed(K, g) for _ in range(3))
    r = bench.roofline(best, K, lg.alg_bytes)
    print(w, lg.name, json.dumps(r), flush=True)
    del lg, g
    torch.cuda.empty_cache()
