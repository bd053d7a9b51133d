#!/usr/bin/env python3
"""Names the samples of `PLUGIN_BENCH_SAMPLE=<Hz> tests/host/plugin_bench ...` (stderr lines "S module+0xoff ...", innermost frame first)
with llvm-symbolizer and prints two tables: CPU time by innermost function (self) and by the first frame inside the plugin / the test
runtime (inclusive by facade).  usage: scripts/walk_profile.py stderr.txt [top [function-whose-callers-to-list]]"""
import collections
import os
import subprocess
import sys

SYM = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def main():
    lines = [ln.split()[1:] for ln in open(sys.argv[1], errors="replace") if ln.startswith("S ")]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    addrs = sorted({a for ln in lines for a in ln if not a.startswith("?")})
    by_mod = collections.defaultdict(list)
    for a in addrs:
        mod, off = a.rsplit("+", 1)
        by_mod[mod].append(off)
    name = {}
    for mod, offs in by_mod.items():
        if not os.path.exists(mod):
            for o in offs:
                name[f"{mod}+{o}"] = os.path.basename(mod)
            continue
        out = subprocess.run([SYM, "--obj=" + mod, "--functions=linkage", "--demangle", "--no-inlines"] + offs,
                             capture_output=True, text=True).stdout.split("\n\n")
        for o, blk in zip(offs, out):
            fn = blk.strip().splitlines()[0] if blk.strip() else "??"
            if fn == "??":
                fn = os.path.basename(mod)
            name[f"{mod}+{o}"] = fn.replace("(anonymous namespace)::", "").split("(")[0][:90]
    n = len(lines)
    self_t = collections.Counter()
    incl = collections.Counter()
    for ln in lines:
        fns = [name.get(a, "?") for a in ln]
        if not fns:
            continue
        self_t[fns[0]] += 1
        for fn in dict.fromkeys(fns):
            incl[fn] += 1
    if len(sys.argv) > 3:  # callers of one function: the chains that end in it
        chains = collections.Counter()
        for ln in lines:
            fns = [name.get(a, "?") for a in ln]
            if sys.argv[3] in fns:
                i = fns.index(sys.argv[3])
                chains[" <- ".join(fns[i:i + 4])] += 1
        for ch, c in chains.most_common(top):
            print(f"{100.0 * c / n:6.2f} %  {ch}")
        return
    print(f"{n} samples")
    print("-- self")
    for fn, c in self_t.most_common(top):
        print(f"{100.0 * c / n:6.2f} %  {fn}")
    print(f"-- inclusive (within the innermost {max(len(l) for l in lines)} frames)")
    for fn, c in incl.most_common(top):
        print(f"{100.0 * c / n:6.2f} %  {fn}")


if __name__ == "__main__":
    main()
