import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "host", "libms2shim.so")  # test-only host runtime (tests/host/ms2shim.c)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared objects are build artefacts (git-ignored): a fresh checkout gets them here, the same way the driver's
    build step does -- hipcc cross-compiles gfx950 without a GPU.  Building is not a fallback: nothing below computes
    without the HIP library."""
    pkg = os.path.join(ROOT, "mediastreamer2_amd")
    need = [os.path.join(pkg, n) for n in ("libmsmi355x.so", "libmsmi355xfilters.so")]
    need.append(os.path.join(ROOT, "tests", "host", "libms2shim.so"))
    need.append(os.path.join(ROOT, "oracle", "liboracle.so"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as entry
        entry.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def ctx():
    """One mi_ctx on device 0 -- fails loudly if the HIP library or GPU is missing."""
    import mediastreamer2_amd as ms
    c = ms.Context(0)
    yield c
    c.close()


def synth_pcm(stream_id, n, sigma=3000.0, tone_hz=1000.0, rate=48000, t0=0):
    """SURVEY 8(d) synthetic audio: N(0,sigma) + -20 dBFS tone, clipped to +-32767."""
    rng = np.random.default_rng(0x5EED + stream_id)
    t = (np.arange(n) + t0) / rate
    x = rng.normal(0.0, sigma, n) + 3276.7 * np.sin(2 * np.pi * tone_hz * t)
    return np.clip(np.round(x), -32767, 32767).astype(np.int16)


@pytest.fixture
def pcm():
    return synth_pcm
