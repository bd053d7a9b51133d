import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["MS2SHIM_VERBOSE"] = "1"
import numpy as np
from test_gpu_plugin import *
h = Host()
ec = h.create(MS_SPEEX_EC_ID)
h.call_int(ec, SET_SAMPLE_RATE, 16000); h.call_int(ec, mid(EC_IFACE, 2, 4), 128)
s_ref, s_mic, k_ref, k_mic = h.source(), h.source(), h.sink(), h.sink()
h.link(s_ref, 0, ec, 0); h.link(s_mic, 0, ec, 1); h.link(ec, 0, k_ref, 0); h.link(ec, 1, k_mic, 0)
h.S.ms_ticker_attach(h.ticker, ec)
x = np.arange(160, dtype=np.int16)
for t in range(6):
    h.push(s_ref, x); h.push(s_mic, x)
for t in range(9):
    h.step(1)
    print("tick", t, "k_ref bytes", h.S.ms2shim_sink_size(k_ref), "k_mic bytes", h.S.ms2shim_sink_size(k_mic), flush=True)
