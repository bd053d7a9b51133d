"""CPU control flow of the parts that need more than one GPU, through the real entry points and the host-memory double of
the kernel library (tests/host/mi_double.cpp, TEST INFRASTRUCTURE -- the product has no CPU path):
  * one process, many devices: the plugin's ticker hubs open their contexts on the least-loaded device of
    MSMI355X_DEVICES (default: every visible one), so the tickers of one mediastreamer2 process spread over the node;
  * the split conference through the C ABI: mi_mixer_partial_sum -> mi_exchange_allreduce_i32 -> mi_mixer_finalize with
    a rank per thread (the double's exchange is an in-process rendezvous with RCCL's contract; its mixer is the mixer's
    definition), bit-equal to the whole conference mixed in one place -- no gloo, no torch."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
DOUBLE = os.path.join(HOST, "double")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", HOST, "all"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


def test_the_double_exports_the_whole_c_abi(built):
    """every symbol include/msmi355x.h declares (the list the product library is checked against as well)"""
    sys.path.insert(0, ROOT)
    from mediastreamer2_amd._lib import EXPORTS
    L = C.CDLL(os.path.join(DOUBLE, "libmsmi355x.so"))
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    assert not missing, missing


HUBS = r"""
import ctypes as C, os, sys
S = C.CDLL(os.path.join({host!r}, "libms2shim.so"), mode=C.RTLD_GLOBAL)
vp = C.c_void_p
for fn in ("ms_factory_new", "ms_factory_create_filter", "ms_ticker_new", "ms2shim_new_source", "ms2shim_new_sink"):
    getattr(S, fn).restype = vp
S.ms_factory_create_filter.argtypes = [vp, C.c_int]
S.ms_factory_load_plugin.argtypes = [vp, C.c_char_p]
S.ms_filter_link.argtypes = [vp, C.c_int, vp, C.c_int]
S.ms_filter_call_method.argtypes = [vp, C.c_uint, vp]
S.ms_ticker_attach.argtypes = [vp, vp]
S.ms_ticker_detach.argtypes = [vp, vp]
S.ms_ticker_step.argtypes = [vp]
S.ms2shim_new_source.argtypes = [vp]
S.ms2shim_new_sink.argtypes = [vp]
S.ms2shim_source_push.argtypes = [vp, vp, C.c_size_t]
S.ms2shim_register_test_filters.argtypes = [vp]
S.ms_filter_destroy.argtypes = [vp]
S.ms_ticker_destroy.argtypes = [vp]
fac = S.ms_factory_new()
S.ms2shim_register_test_filters(fac)
plugin = os.path.join({double!r}, "libmsmi355xfilters.so")
assert S.ms_factory_load_plugin(fac, plugin.encode()) == 0
P = C.CDLL(plugin)
mid = lambda fid, idx, size: ((fid & 0xFFFF) << 16) | (idx << 8) | (size & 0xFF)
block = (C.c_int16 * 160)(*range(160))
graphs = []
for t in range(int(sys.argv[1])):
    tk = S.ms_ticker_new()
    src, vol, snk = S.ms2shim_new_source(fac), S.ms_factory_create_filter(fac, 43), S.ms2shim_new_sink(fac)
    r = C.c_int(16000)
    assert S.ms_filter_call_method(vol, mid(2, 0, 4), C.byref(r)) == 0
    S.ms_filter_link(src, 0, vol, 0)
    S.ms_filter_link(vol, 0, snk, 0)
    S.ms_ticker_attach(tk, src)
    S.ms2shim_source_push(src, block, 320)
    S.ms_ticker_step(tk)
    S.ms_ticker_step(tk)
    graphs.append((tk, src, vol, snk))
dev = (C.c_int * 64)()
n = P.ms_mi355x_hub_devices(dev, 64)
print("devices", sorted(dev[i] for i in range(n)))
for tk, src, vol, snk in graphs[::2]:  # every other call ends: its hub, context and device share go
    S.ms_ticker_detach(tk, src)
    for f in (src, vol, snk):
        S.ms_filter_destroy(f)
    S.ms_ticker_destroy(tk)
n = P.ms_mi355x_hub_devices(dev, 64)
print("after", sorted(dev[i] for i in range(n)))
"""


@pytest.mark.parametrize("env,tickers,want,after", [
    ({"MSMI355X_DOUBLE_DEVICES": "4"}, 8, [0, 0, 1, 1, 2, 2, 3, 3], 4),                                # default: every visible device
    ({"MSMI355X_DOUBLE_DEVICES": "8", "MSMI355X_DEVICES": "1,5"}, 6, [1, 1, 1, 5, 5, 5], 3),          # an explicit list
    ({"MSMI355X_DOUBLE_DEVICES": "8", "MSMI355X_DEVICE": "6"}, 3, [6, 6, 6], 1),                      # the older single-device switch
])
def test_ticker_hubs_spread_over_the_devices_of_one_process(built, env, tickers, want, after):
    e = {k: v for k, v in os.environ.items() if not k.startswith("MSMI355X_")}
    e.update(env)
    r = subprocess.run([sys.executable, "-c", HUBS.format(host=HOST, double=DOUBLE), str(tickers)], capture_output=True, text=True,
                       timeout=120, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if ln.startswith(("devices", "after")))
    assert eval(lines["devices"]) == want
    assert len(eval(lines["after"])) == after


@pytest.mark.parametrize("world", [2, 4, 8])
def test_split_conference_through_the_c_entry_points_with_a_rank_per_thread(built, world):
    """(8 ranks on 8 devices: the shape of the driver's 8-GPU run of the exchange step, one thread and one device per rank)"""
    L = C.CDLL(os.path.join(DOUBLE, "libmsmi355x.so"))
    vp, pp = C.c_void_p, C.POINTER(C.c_void_p)
    L.mi_ctx_create.argtypes = [C.c_int, vp, pp]
    L.mi_mixer_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, pp]
    L.mi_mixer_partial_sum.argtypes = [vp, vp, vp, vp]
    L.mi_mixer_finalize.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    L.mi_mixer_process.argtypes = [vp, vp, vp, C.c_int, vp]
    L.mi_exchange_unique_id.argtypes = [vp, C.c_size_t]
    L.mi_exchange_create.argtypes = [vp, C.c_int, C.c_int, vp, pp]
    L.mi_exchange_allreduce_i32.argtypes = [vp, vp, C.c_size_t]
    L.mi_exchange_destroy.argtypes = [vp]
    L.mi_last_error.restype = C.c_char_p
    os.environ["MSMI355X_DOUBLE_DEVICES"] = "8"
    nconf, mm, ns, ticks = 6, 32, 480, 5
    per = mm // world
    rng = np.random.default_rng(11)
    x = rng.integers(-16000, 16000, (ticks, nconf, mm, ns), dtype=np.int16)
    uid = (C.c_ubyte * 128)()
    assert L.mi_exchange_unique_id(uid, 128) == 0
    outs, errs = [None] * world, []

    def rank(r):
        try:
            ctx, mx, ex = vp(), vp(), vp()
            assert L.mi_ctx_create(r, None, C.byref(ctx)) == 0, L.mi_last_error()
            assert L.mi_mixer_create(ctx, nconf, mm // world, ns, C.byref(mx)) == 0
            assert L.mi_exchange_create(ctx, world, r, uid, C.byref(ex)) == 0, L.mi_last_error()   # returns when all four are in
            got = []
            for t in range(ticks):
                mine = np.ascontiguousarray(x[t, :, r * per:(r + 1) * per])
                total = np.zeros((nconf, ns), np.int32)
                out = np.zeros_like(mine)
                assert L.mi_mixer_partial_sum(mx, mine.ctypes.data, None, total.ctypes.data) == 0
                assert L.mi_exchange_allreduce_i32(ex, total.ctypes.data, total.size) == 0
                assert L.mi_mixer_finalize(mx, mine.ctypes.data, None, total.ctypes.data, 1, out.ctypes.data) == 0
                got.append(out)
            outs[r] = np.stack(got)
            L.mi_exchange_destroy(ex)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errs, errs
    ctx, whole = vp(), vp()
    assert L.mi_ctx_create(0, None, C.byref(ctx)) == 0 and L.mi_mixer_create(ctx, nconf, mm, ns, C.byref(whole)) == 0
    for t in range(ticks):
        ref = np.zeros((nconf, mm, ns), np.int16)
        assert L.mi_mixer_process(whole, np.ascontiguousarray(x[t]).ctypes.data, None, 1, ref.ctypes.data) == 0
        want = (x[t].astype(np.int64).sum(1, keepdims=True) - x[t]).clip(-32767, 32767)   # audiomixer.c:33-51,:301-344
        np.testing.assert_array_equal(ref, want)
        for r in range(world):
            np.testing.assert_array_equal(outs[r][t], ref[:, r * per:(r + 1) * per])
    bad = vp()
    assert L.mi_exchange_create(ctx, 2, 9, uid, C.byref(bad)) != 0   # rank outside the communicator
