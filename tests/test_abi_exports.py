"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/msmi355x.h declares; without a GPU, create calls fail
loudly (no CPU fallback)."""
import os
import re
import subprocess

import pytest

import mediastreamer2_amd as ms
from mediastreamer2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "msmi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_header_and_export_list_agree():
    assert _header_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    missing = [s for s in _header_symbols() if not hasattr(L, s)]
    assert not missing, f"libmsmi355x.so lacks {missing}"
    assert L.mi_abi_version() == 3


def test_library_is_in_tree_and_has_gfx950_code():
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "mediastreamer2_amd")
    out = subprocess.run(["strings", "-a", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ms.MiError) as e:
        ms.Context(0)
    assert e.value.code == _lib.MI_ENODEV


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under mediastreamer2_amd/ or include/ may name it."""
    bad = []
    for base in ("mediastreamer2_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp", ".cc", ".inl")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    if re.search(r"\bimport oracle\b|from oracle\b|ms2_oracle\.h|liboracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_plugin_exports_every_declared_descriptor():
    """include/ms2_plugin_abi.h declares the plugin's entry point and one MSFilterDesc per facade: the built
    libmsmi355xfilters.so must define them all (checked with nm: loading it needs a mediastreamer2 runtime)."""
    hdr = open(os.path.join(ROOT, "include", "ms2_plugin_abi.h")).read()
    declared = set(re.findall(r"extern\s+MS(?:Filter|Scaler)Desc\s+(ms_mi355x_[a-z0-9_]+)\s*;", hdr))
    assert len(declared) >= 17
    so = os.path.join(ROOT, "mediastreamer2_amd", "libmsmi355xfilters.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r"\b(ms_mi355x_[a-z0-9_]+|libmsmi355xfilters_init|ms_mi355x_flush|ms_mi355x_shutdown)\b", out))
    assert declared <= defined, sorted(declared - defined)
    assert {"libmsmi355xfilters_init", "ms_mi355x_flush", "ms_mi355x_shutdown"} <= defined
    # and it resolves its kernels from libmsmi355x.so only: no oracle, no CPU implementation linked in
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "libmsmi355x.so" in needed and "liboracle" not in needed


def _cc(args, **kw):
    return subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include")] + args,
                          capture_output=True, text=True, **kw)


def test_headers_are_plain_c_and_the_example_builds(tmp_path):
    """The boundary is a C ABI: both headers compile as C99, and examples/conference_bridge.c (the session API in a
    server's main loop) builds against libmsmi355x.so alone; without a GPU it exits loudly instead of falling back."""
    probe = tmp_path / "hdr.c"
    probe.write_text('#include "msmi355x.h"\n#include "ms2_plugin_abi.h"\n'
                     "int main(void) { return mi_abi_version() == MSMI355X_ABI_VERSION ? 0 : 1; }\n")
    r = _cc(["-Werror", "-c", str(probe), "-o", str(tmp_path / "hdr.o")])
    assert r.returncode == 0, r.stderr
    exe = tmp_path / "bridge"
    pkg = os.path.join(ROOT, "mediastreamer2_amd")
    r = _cc(["-Werror", os.path.join(ROOT, "examples", "conference_bridge.c"), "-L", pkg, "-lmsmi355x",
             f"-Wl,-rpath,{pkg}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([str(exe)], capture_output=True, text=True)
        assert run.returncode == 1 and "no CPU fallback" in run.stderr


@pytest.mark.gpu
def test_conference_bridge_example_runs(tmp_path):
    """300 ticks of a 2048-leg G.711 bridge (decode, PLC, resample, AEC, AGC, mix, resample, encode) through the plain-C
    example."""
    exe = tmp_path / "bridge"
    pkg = os.path.join(ROOT, "mediastreamer2_amd")
    r = _cc([os.path.join(ROOT, "examples", "conference_bridge.c"), "-L", pkg, "-lmsmi355x", f"-Wl,-rpath,{pkg}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stderr


def test_split_conference_example_builds(tmp_path):
    """examples/split_conference.c (one process, a thread per GPU, the conference exchange on RCCL through the C ABI) is plain
    C99 against libmsmi355x.so + pthread; without a GPU it exits loudly."""
    exe = tmp_path / "split"
    pkg = os.path.join(ROOT, "mediastreamer2_amd")
    r = _cc(["-Werror", "-D_POSIX_C_SOURCE=200809L", os.path.join(ROOT, "examples", "split_conference.c"), "-L", pkg, "-lmsmi355x", "-lpthread",
             f"-Wl,-rpath,{pkg}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([str(exe)], capture_output=True, text=True)
        assert run.returncode == 1 and "no CPU fallback" in run.stderr


@pytest.mark.gpu
def test_split_conference_example_runs_on_every_visible_gpu(tmp_path):
    """64 conferences x 32 members split over the visible GPUs (one thread and one exchange rank per GPU; a single rank on a
    one-GPU box: RCCL is still initialised and its all-reduce still sits between partial_sum and finalize), 20 ticks, every
    member's mix compared with the whole-conference mix."""
    exe = tmp_path / "split"
    pkg = os.path.join(ROOT, "mediastreamer2_amd")
    r = _cc(["-D_POSIX_C_SOURCE=200809L", os.path.join(ROOT, "examples", "split_conference.c"), "-L", pkg, "-lmsmi355x", "-lpthread",
             f"-Wl,-rpath,{pkg}", "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    last = run.stdout.strip().splitlines()[-1].split()  # (RCCL prints its version banner to stdout first)
    assert run.returncode == 0 and last[0] == "ok", run.stderr
    import torch
    assert int(last[1]) == max(k for k in (1, 2, 4, 8, 16, 32) if k <= torch.cuda.device_count())


def test_stub_ticker_layout_follows_the_reference(tmp_path):
    """include/mediastreamer2/msticker.h:73-98: lock, cond, two list pointers, thread, then interval / exec_id / ticks /
    time.  The stub in ms2_plugin_abi.h must put the three fields a filter reads (msfilter.h:203: f->ticker->time,
    ->interval; ->ticks) where a build against the real headers finds them -- on x86-64 glibc (40-byte mutex, 48-byte
    condition variable) that is 112 / 120 / 128."""
    src = tmp_path / "tk.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include <pthread.h>\n#include "ms2_plugin_abi.h"\n'
                   "struct ref_ticker { pthread_mutex_t lock; pthread_cond_t cond; void *execution_list; void *task_list;\n"
                   "  pthread_t thread; int interval; int exec_id; unsigned int ticks; unsigned long long time; };\n"
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(MSTicker, interval), offsetof(MSTicker, ticks),\n'
                   "  offsetof(MSTicker, time), offsetof(struct ref_ticker, interval), offsetof(struct ref_ticker, ticks),\n"
                   "  offsetof(struct ref_ticker, time)); return 0; }\n")
    exe = tmp_path / "tk"
    r = _cc([str(src), "-o", str(exe)])
    assert r.returncode == 0, r.stderr
    v = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert v[:3] == v[3:], v
    assert v[:3] == [112, 120, 128]
