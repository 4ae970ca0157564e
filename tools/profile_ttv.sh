#!/bin/bash
# rocprofv3 kernel stats of tools/profile_ttv.py (C2 with and without timing tables)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/ttv_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python $R/tools/profile_ttv.py > $out/run.json 2> $out/run.err
python - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)
out = ["# rocprofv3 --kernel-trace --stats -- python tools/profile_ttv.py"]
for r in csv.DictReader(open(f[0])):
    if "transit" in r["Name"]:
        out.append("%-110s calls %4s  avg %9.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
open("$out/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
