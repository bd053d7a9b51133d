"""The two hardware idioms the canceller's arithmetic rests on, checked on their own against the host:
scripts/micro/dpp_chain.hip -- a float sum over the 64 lanes in LANE ORDER as a systolic v_add_f32_dpp wave_shr:1 chain equals
the sequential host loop bit for bit (and the v_readlane form it replaced); scripts/micro/pk_hazard.hip -- a packed-FP32 write
followed at once by a 32-bit read of one half of the pair returns the right value (inline asm is opaque to the compiler's
hazard recogniser, which puts a wait state there in its own code)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_and_run(name, tmp_path):
    exe = tmp_path / name
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", str(exe),
                        os.path.join(ROOT, "scripts", "micro", name + ".hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    return run.stdout


def test_dpp_chain_equals_the_sequential_sum(tmp_path):
    out = build_and_run("dpp_chain", tmp_path)
    lines = [ln for ln in out.splitlines() if "bit-exact" in ln]
    assert len(lines) == 2 and all("bit-exact yes" in ln for ln in lines), out


def test_packed_write_then_half_read_needs_no_wait_state(tmp_path):
    out = build_and_run("pk_hazard", tmp_path)
    lines = [ln for ln in out.splitlines() if "mismatches" in ln]
    assert len(lines) == 2 and all(ln.split(":")[1].split()[0] == "0" for ln in lines), out
