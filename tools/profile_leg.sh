#!/bin/bash
# rocprofv3 --kernel-trace --stats of one eager bench step: bash tools/profile_leg.sh <name> <bench.py arguments...>
#   -> gpurun_out/r05/<name>_kernel_stats.csv (+ a sorted text table on stdout)
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
out=$R/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-extras --no-stats --no-graph > $out/${name}_trace.log 2>&1
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${name}_kernel_stats.csv
python - "$out/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s %6s calls %10.1f us avg %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
