#!/bin/bash
# Resource summary of the canceller's tick kernels (what tests/test_kernel_resources_cpu.py checks), from a device-only
# compile of aec.hip: code bytes, VGPRs, spills, LDS.  Usage: scripts/kres.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/../mediastreamer2_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" -c aec.hip -o $T/aec_dev.o 2> $T/remarks.txt || { tail -30 $T/remarks.txt; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/aec_dev.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$T/aec_gfx950.o 2>/dev/null || cp $T/aec_dev.o $T/aec_gfx950.o
python3 - "$T" <<'PY'
import re, subprocess, sys
t = sys.argv[1]
rem = open(t + "/remarks.txt").read()
syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", t + "/aec_gfx950.o"], capture_output=True, text=True).stdout
sizes = {ln.split()[7]: int(ln.split()[2]) for ln in syms.splitlines() if " FUNC " in ln and "aec_tick_kernel" in ln}
for b in re.split(r"remark: Function Name: ", rem)[1:]:
    name = b.split()[0]
    if "aec_tick_kernel" not in name: continue
    u = {m.group(1).strip(): m.group(2).strip() for m in re.finditer(r"remark:\s+([A-Za-z /\[\]]+):\s+(\S+)", b)}
    name_short = name.replace("_ZN12_GLOBAL__N_115aec_tick_kernel", "tick")
    print(name_short, "code", sizes.get(name), "VGPRs", u.get("VGPRs"), "spill", u.get("VGPRs Spill"), "SGPRs", u.get("TotalSGPRs"), "sspill", u.get("SGPRs Spill"), "LDS", u.get("LDS Size [bytes/block]"), "occ", u.get("Occupancy [waves/SIMD]"))
PY
rm -rf $T
