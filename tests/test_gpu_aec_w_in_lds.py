"""The W-in-LDS experiment (csrc/aec_tick.hpp: WLDS, MSMI355X_AEC_W_IN_LDS=1; profiles/r05_w_in_lds.txt) computes what the product
form computes: the chained path over 256 staggered legs and 400 ticks of an echo scene with double talk (foreground updates and
background resets on either frame of a tick), every mix of every tick hashed -- the same hash with the switch on and off."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(on):
    env = dict(os.environ)
    env.pop("MSMI355X_AEC_W_IN_LDS", None)
    if on:
        env["MSMI355X_AEC_W_IN_LDS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r05_wlds_check.py"), "400"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("wlds-check")][-1].split()
    return line[1], line[3]


def test_w_in_lds_is_bit_equal_to_the_product_form():
    (f0, h0), (f1, h1) = run(False), run(True)
    assert (f0, f1) == ("0", "1") and h0 == h1
