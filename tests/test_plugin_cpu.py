"""CPU-side checks of the plugin boundary (no GPU, no compute): the plugin shared object loads
through the factory's loader convention, registers descriptors that take over the reference's
ids, and its method tables answer like the reference's for everything that is host-only."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mediastreamer2_amd")


def mid(fid, idx, argsize):
    return ((fid & 0xFFFF) << 16) | (idx << 8) | (argsize & 0xFF)


@pytest.fixture(scope="module")
def shim():
    for lib in ("libmsmi355xfilters.so", "libmsmi355x.so"):
        assert os.path.exists(os.path.join(PKG, lib)), f"{lib} not built (run __graft_entry__.build())"
    assert os.path.exists(os.path.join(ROOT, "tests", "host", "libms2shim.so")), "tests/host/libms2shim.so not built"
    S = C.CDLL(os.path.join(ROOT, "tests", "host", "libms2shim.so"), mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    S.ms_factory_new.restype = vp
    S.ms_factory_create_filter.restype = vp
    S.ms_factory_create_filter.argtypes = [vp, C.c_int]
    S.ms_factory_load_plugin.argtypes = [vp, C.c_char_p]
    S.ms2shim_filter_name.restype = C.c_char_p
    S.ms2shim_filter_name.argtypes = [vp]
    S.ms2shim_filter_flags.restype = C.c_uint
    S.ms2shim_filter_flags.argtypes = [vp]
    S.ms_filter_call_method.argtypes = [vp, C.c_uint, vp]
    S.ms_filter_destroy.argtypes = [vp]
    S.ms2shim_method_id.restype = C.c_uint
    fac = S.ms_factory_new()
    # without a HIP device the plugin registers nothing (the reference's filters stay in charge); this module inspects
    # the descriptors on a CPU-only box, so it asks for registration all the same
    os.environ["MSMI355X_REGISTER_WITHOUT_DEVICE"] = "1"
    assert S.ms_factory_load_plugin(fac, os.path.join(PKG, "libmsmi355xfilters.so").encode()) == 0
    return S, fac


def test_plugin_exports_the_loader_entry_point():
    """src/base/msfactory.c:549-555: "<file name up to .so>_init"."""
    P = C.CDLL(os.path.join(ROOT, "tests", "host", "libms2shim.so"), mode=C.RTLD_GLOBAL)  # provides the ms2 symbols the plugin needs
    L = C.CDLL(os.path.join(PKG, "libmsmi355xfilters.so"))
    assert hasattr(L, "libmsmi355xfilters_init")
    for d in ("ms_mi355x_resample_desc", "ms_mi355x_audio_mixer_desc", "ms_mi355x_volume_desc",
              "ms_mi355x_equalizer_desc", "ms_mi355x_speex_ec_desc"):
        assert hasattr(L, d)
    del P


@pytest.mark.parametrize("fid,name,nin,pump", [(41, b"MSResample", 1, 0), (43, b"MSVolume", 1, 0),
                                              (61, b"MSEqualizer", 1, 0), (68, b"MSAudioMixer", 50, 1),
                                              (28, b"MSSpeexEC", 2, 0)])
def test_descriptors_take_over_reference_ids(shim, fid, name, nin, pump):
    S, fac = shim
    f = S.ms_factory_create_filter(fac, fid)
    assert f and S.ms2shim_filter_name(f) == name
    flags = S.ms2shim_filter_flags(f)
    assert flags & 2, "MS_FILTER_IS_HW_ACCELERATED"
    assert bool(flags & 1) == bool(pump), "MS_FILTER_IS_PUMP only on the mixer (audiomixer.c:464)"
    S.ms_filter_destroy(f)


def test_method_tables_host_side(shim):
    S, fac = shim
    assert S.ms2shim_method_id(43, 2, 4) == mid(43, 2, 4)
    vol = S.ms_factory_create_filter(fac, 43)
    g = C.c_float(0.25)
    assert S.ms_filter_call_method(vol, mid(43, 2, 4), C.byref(g)) == 0       # MS_VOLUME_SET_GAIN
    out = C.c_float()
    assert S.ms_filter_call_method(vol, mid(43, 14, 4), C.byref(out)) == 0    # MS_VOLUME_GET_GAIN
    assert out.value == 0.25
    db = C.c_float(3.0)
    assert S.ms_filter_call_method(vol, mid(43, 13, 4), C.byref(db)) == 0     # SET_DB_GAIN: 10^(dB/10) (A10)
    S.ms_filter_call_method(vol, mid(43, 14, 4), C.byref(out))
    assert abs(out.value - 10 ** 0.3) < 1e-6
    bad = C.c_float(2.0)
    assert S.ms_filter_call_method(vol, mid(43, 5, 4), C.byref(bad)) == -1    # EA threshold range check (:305-314)
    assert S.ms_filter_call_method(vol, mid(2, 3, 4), C.byref(bad)) == -1     # unknown base method: silent -1
    S.ms_filter_destroy(vol)

    class Ctl(C.Structure):
        _fields_ = [("pin", C.c_int), ("v", C.c_float)]
    mx = S.ms_factory_create_filter(fac, 68)
    assert S.ms_filter_call_method(mx, mid(68, 0, 8), C.byref(Ctl(50, 1.0))) == -1  # pin out of range
    assert S.ms_filter_call_method(mx, mid(68, 0, 8), C.byref(Ctl(49, 0.5))) == 0
    r = C.c_int(16000)
    assert S.ms_filter_call_method(mx, mid(2, 0, 4), C.byref(r)) == 0
    q = C.c_int()
    assert S.ms_filter_call_method(mx, mid(2, 1, 4), C.byref(q)) == 0 and q.value == 16000
    S.ms_filter_destroy(mx)

    ec = S.ms_factory_create_filter(fac, 28)
    d = C.c_int(40)
    assert S.ms_filter_call_method(ec, mid(16388, 0, 4), C.byref(d)) == 0     # MS_ECHO_CANCELLER_SET_DELAY
    g2 = C.c_int()
    assert S.ms_filter_call_method(ec, mid(16388, 7, 4), C.byref(g2)) == 0 and g2.value == 40
    b = C.c_ubyte(1)
    assert S.ms_filter_call_method(ec, mid(16388, 3, 1), C.byref(b)) == 0     # bypass mode
    S.ms_filter_destroy(ec)


def test_without_a_device_the_plugin_steps_aside():
    """No HIP device and no override: libmsmi355xfilters_init registers nothing, so the factory keeps handing out whatever
    was registered before it (the reference's CPU filters in a real host; nothing at all in the test runtime) -- run in a
    child process so that the override of the fixture above does not apply."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, os, sys\n"
        "os.environ.pop('MSMI355X_REGISTER_WITHOUT_DEVICE', None)\n"
        "os.environ['HIP_VISIBLE_DEVICES'] = ''\n"
        f"S = C.CDLL({os.path.join(ROOT, 'tests', 'host', 'libms2shim.so')!r}, mode=C.RTLD_GLOBAL)\n"
        "S.ms_factory_new.restype = C.c_void_p\n"
        "S.ms_factory_load_plugin.argtypes = [C.c_void_p, C.c_char_p]\n"
        "S.ms_factory_lookup_filter_by_id.restype = C.c_void_p\n"
        "S.ms_factory_lookup_filter_by_id.argtypes = [C.c_void_p, C.c_int]\n"
        "fac = S.ms_factory_new()\n"
        f"assert S.ms_factory_load_plugin(fac, {os.path.join(PKG, 'libmsmi355xfilters.so')!r}.encode()) == 0\n"
        "sys.exit(0 if not S.ms_factory_lookup_filter_by_id(fac, 41) else 7)\n"
    )
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "NOT registered" in p.stderr


@pytest.mark.parametrize("claim", [False, True])
def test_webrtc_aec_name_is_claimed_only_on_request(claim):
    """The factory's default echo-canceller NAME is "MSWebRTCAEC" (msfactory.c:245); audio_stream_new_with_sessions looks the
    name up and falls back to MS_SPEEX_EC_ID (audiostream.c:2128-2158).  By default the plugin does not answer to that name;
    with MSMI355X_CLAIM_WEBRTC_AEC=1 it does, with a descriptor whose text says it is not AEC3 and whose methods are the
    speex-class canceller's."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, os, sys\n"
        "os.environ['MSMI355X_REGISTER_WITHOUT_DEVICE'] = '1'\n"
        + ("os.environ['MSMI355X_CLAIM_WEBRTC_AEC'] = '1'\n" if claim else "os.environ.pop('MSMI355X_CLAIM_WEBRTC_AEC', None)\n") +
        f"S = C.CDLL({os.path.join(ROOT, 'tests', 'host', 'libms2shim.so')!r}, mode=C.RTLD_GLOBAL)\n"
        "S.ms_factory_new.restype = C.c_void_p\n"
        "S.ms_factory_load_plugin.argtypes = [C.c_void_p, C.c_char_p]\n"
        "S.ms_factory_lookup_filter_by_name.restype = C.c_void_p\n"
        "S.ms_factory_lookup_filter_by_name.argtypes = [C.c_void_p, C.c_char_p]\n"
        "fac = S.ms_factory_new()\n"
        f"assert S.ms_factory_load_plugin(fac, {os.path.join(PKG, 'libmsmi355xfilters.so')!r}.encode()) == 0\n"
        "d = S.ms_factory_lookup_filter_by_name(fac, b'MSWebRTCAEC')\n"
        "assert S.ms_factory_lookup_filter_by_name(fac, b'MSSpeexEC')\n"
        "if d:\n"
        "    text = C.cast(C.c_void_p.from_address(d + 16).value, C.c_char_p).value\n"  # MSFilterDesc.text (msfilter.h:161-178)
        "    assert b'NOT WebRTC AEC3' in text, text\n"
        "sys.exit(1 if d else 0)\n"
    )
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == (1 if claim else 0), p.stdout + p.stderr
    assert ("this is not AEC3" in p.stderr) == claim
