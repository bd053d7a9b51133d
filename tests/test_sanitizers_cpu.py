"""The plugin's threaded host runtime (per-ticker hubs with their own lock, a registry under a shared mutex, banks that
grow, empty and die, hubs retired by whichever scope drops the last reference) under ThreadSanitizer and AddressSanitizer
+ UBSan, on the CPU box: the SAME sources (mediastreamer2_amd/host/filters.cpp, built with
`make -C mediastreamer2_amd/host SAN=...`), linked against the host-memory double of the kernel library
(tests/host/mi_double.cpp) and driven by tests/host/san_stress.c -- three threads running 'calls' on tickers of their own
(resample + volume chains, echo cancellers, three-party conferences, calls that end mid-way), one growing a ticker to 150
filters and back, one walking every hub all the while.  A clean report is the test; the lifetime bug ThreadSanitizer found
while this was written (a flag read after the reference was dropped) is described at TickerHub::life in filters.cpp.
GPU sanitizers are not available on the pool; the kernels have their parity tests instead."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", HOST, "-j4", "san"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("variant,marks", [("san-thread", ("ThreadSanitizer",)),
                                           ("san-asan", ("AddressSanitizer", "LeakSanitizer", "runtime error"))])
def test_host_runtime_is_clean_under_the_sanitizers(built, variant, marks):
    exe = os.path.join(HOST, variant, "san_stress")
    plugin = os.path.join(HOST, variant, "libmsmi355xfilters.so")
    env = dict(os.environ, MSMI355X_DOUBLE_DEVICES="4", TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1",
               ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    for rep in range(3):  # the interleavings differ from run to run
        r = subprocess.run([exe, plugin, "12"], capture_output=True, text=True, timeout=600, env=env)
        report = [ln for ln in r.stderr.splitlines() if any(m in ln for m in marks)]
        assert r.returncode == 0 and not report, "\n".join(report[:20]) + r.stderr[-1500:]
        assert r.stdout.startswith("ok ")
