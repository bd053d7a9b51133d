"""The canceller's tick kernel lives on three resource edges at once; a change that pushes it over any of them costs
performance silently, so the build is checked here (hipcc cross-compiles without a GPU):
  * 256 VGPRs = two waves per SIMD; not one register spilled to scratch at 48 kHz (F = 256);
  * at most 20 KB of LDS per wave = eight waves per CU;
  * under 64 KB of code: the instruction cache two CUs share (99.99 % hits measured at that size)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mediastreamer2_amd", "csrc", "aec.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def build(tmp_path_factory):
    d = tmp_path_factory.mktemp("aec_res")
    obj = d / "aec_dev.o"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", str(obj)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return d, obj, r.stderr


def usage(remarks, kernel_substr):
    """{field: value} of the kernel-resource-usage remark block of the first function whose name contains kernel_substr"""
    blocks = re.split(r"remark: Function Name: ", remarks)
    for b in blocks[1:]:
        if kernel_substr in b.split()[0]:
            return {m.group(1).strip(): m.group(2).strip() for m in re.finditer(r"remark:\s+([A-Za-z /\[\]]+):\s+(\S+)", b)}
    raise AssertionError(f"no remarks for {kernel_substr}")


def test_tick_kernel_registers_and_lds(build):
    _, _, remarks = build
    u = usage(remarks, "aec_tick_kernelILi256E")
    assert int(u["VGPRs"]) <= 256 and int(u["VGPRs Spill"]) == 0, u
    assert int(u["Occupancy [waves/SIMD]"]) == 2, u
    assert int(u["LDS Size [bytes/block]"]) <= 20480, u  # one wave per block: 8 per CU
    for k in ("aec_tick_kernelILi128E", "aec_tick_kernelILi64E"):
        v = usage(remarks, k)
        assert int(v["VGPRs Spill"]) <= 8, (k, v)  # the small frames run at 3 / 4 waves per SIMD: a handful of spills is the price


def test_tick_kernel_code_fits_the_instruction_cache(build):
    d, obj, _ = build
    dev = d / "aec_gfx950.o"
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={obj}",
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], capture_output=True, text=True)
    if r.returncode != 0 or not dev.exists():  # --cuda-device-only may already emit the bare code object
        dev = obj
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", str(dev)], capture_output=True, text=True).stdout
    sizes = {ln.split()[7]: int(ln.split()[2]) for ln in syms.splitlines() if " FUNC " in ln and "aec_tick_kernel" in ln}
    big = [v for k, v in sizes.items() if "ILi256E" in k]
    assert big, syms[-2000:]
    assert max(big) < 65536, f"aec_tick_kernel<256> is {max(big)} bytes: over the 64 KB instruction cache"
