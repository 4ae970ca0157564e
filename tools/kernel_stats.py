#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel in a hipcc -save-temps .s file (gfx950):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -save-temps -c exoplanet_amd/csrc/X.hip -o /tmp/x.o
    python tools/kernel_stats.py X-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]
waves/SIMD = 512 // (vgpr + agpr rounded up to 8), at most 8."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(.*", "", o.replace("(anonymous namespace)::", "").replace("void ", "")) for o in out]
    except OSError:
        return names


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        body = m.group(2)

        def g(k):
            r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
            return int(r.group(1)) if r and r.group(1).isdigit() else -1

        rows.append((m.group(1), g("next_free_vgpr"), g("accum_offset"), g("next_free_sgpr"),
                     g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = demangle([r[0] for r in rows])
    for (raw, tot, acc, sg, scr, lds), nm in zip(rows, names):
        if flt and flt not in nm:
            continue
        waves = min(8, 512 // max(8, (tot + 7) // 8 * 8))
        print(f"{nm[:110]:110s} regs={tot:4d} (arch {acc:3d}) sgpr={sg:4d} scratch={scr:5d} lds={lds:6d} waves/SIMD={waves}")


if __name__ == "__main__":
    main()
