"""registers / scratch / occupancy of the kernels of one translation unit: python tools/kres.py exo_celerite.hip [filter] [-D...]"""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", R + "/include", "-mllvm",
       "-disable-machine-licm", "-c", R + "/exoplanet_amd/csrc/" + src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    short = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
    if pat and pat not in short:
        continue
    print("%-62s VGPR %4s AGPR %4s scratch %5s occ %s LDS %s" % (short[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                               r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
