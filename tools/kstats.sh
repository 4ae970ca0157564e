#!/bin/bash
# per-kernel rocprofv3 averages of one eager bench step: tools/kstats.sh <name> <bench.py arguments ...>
#   -> gpurun_out/<name>_kernel_stats.csv (and the top of it on stdout)
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o p -- python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-stats --no-graph > /tmp/ks_$name.log 2>&1
f=$(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
cp $f $R/gpurun_out/${name}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
