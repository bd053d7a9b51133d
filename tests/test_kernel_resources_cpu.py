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


def usages(remarks, kernel_substr):
    """[(function name, {field: value})] of every kernel-resource-usage remark block whose function name contains kernel_substr"""
    out = []
    for b in re.split(r"remark: Function Name: ", remarks)[1:]:
        name = b.split()[0]
        if kernel_substr in name:
            out.append((name, {m.group(1).strip(): m.group(2).strip() for m in re.finditer(r"remark:\s+([A-Za-z /\[\]]+):\s+(\S+)", b)}))
    assert out, f"no remarks for {kernel_substr}"
    return out


def test_tick_kernel_registers_and_lds(build):
    """every form of the tick kernel (rows / FIFOs / FIFOs + folded resampler: the MODE template parameter)"""
    _, _, remarks = build
    big = list(usages(remarks, "aec_tick_kernelILi256E"))
    assert len(big) == 3, [n for n, _ in big]
    for name, u in big:
        assert int(u["VGPRs"]) <= 256 and int(u["VGPRs Spill"]) == 0, (name, u)
        assert int(u["Occupancy [waves/SIMD]"]) == 2, (name, u)
        assert int(u["LDS Size [bytes/block]"]) <= 20480, (name, u)  # one wave per block: 8 per CU
    for k in ("aec_tick_kernelILi128E", "aec_tick_kernelILi64E"):
        for name, v in usages(remarks, k):
            assert int(v["VGPRs Spill"]) <= 12, (name, v)  # the small frames run at 3 / 4 waves per SIMD: a handful of spills is the price


def test_tick_kernel_code_fits_the_instruction_cache(build):
    d, obj, _ = build
    dev = d / "aec_gfx950.o"
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={obj}",
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", f"--output={dev}"], capture_output=True, text=True)
    if r.returncode != 0 or not dev.exists():  # --cuda-device-only may already emit the bare code object
        dev = obj
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", str(dev)], capture_output=True, text=True).stdout
    sizes = {ln.split()[7]: int(ln.split()[2]) for ln in syms.splitlines() if " FUNC " in ln and "aec_tick_kernel" in ln}
    big = {k: v for k, v in sizes.items() if "ILi256E" in k}
    assert len(big) == 3, syms[-2000:]
    assert max(big.values()) < 65536, f"a form of aec_tick_kernel<256> is over the 64 KB instruction cache: {big}"
    headline = [v for k, v in big.items() if "ILi256ELi2E" in k]  # FIFOs + folded resampler: what the headline and the plugin's fused chain launch
    assert headline and headline[0] <= 64 * 1024 - 1536, f"the headline's form has less than 1.5 KB of instruction cache to spare: {headline}"


def test_group_kernels_registers_lds_and_code(build):
    """the small-frame cancellers with several legs per wavefront (aec_group.hpp): the per-leg scalars live in registers, so
    the kernels sit close to the 256 of two waves per SIMD -- nothing may spill to scratch; LDS for eight waves per CU; code
    inside the instruction cache"""
    d, obj, remarks = build
    for k in ("aec_group_kernelILi128E", "aec_group_kernelILi64E"):
        (name, u), = usages(remarks, k)
        assert int(u["VGPRs"]) <= 256 and int(u["VGPRs Spill"]) == 0 and int(u["ScratchSize [bytes/lane]"]) == 0, (name, u)
        assert int(u["Occupancy [waves/SIMD]"]) >= 2, (name, u)
        assert int(u["LDS Size [bytes/block]"]) <= 20480, (name, u)
    dev = d / "aec_gfx950.o"
    if not dev.exists():
        dev = obj
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", str(dev)], capture_output=True, text=True).stdout
    sizes = {ln.split()[7]: int(ln.split()[2]) for ln in syms.splitlines() if " FUNC " in ln and "aec_group_kernel" in ln}
    assert len(sizes) == 2 and max(sizes.values()) < 65536, sizes
