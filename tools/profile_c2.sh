#!/bin/bash
# rocprofv3 kernel stats of the C2 sweep (bench.py, eager launches so that every kernel is a dispatch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-c2_prof}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stats --no-graph > $out/bench_traced.json 2> $out/bench_traced.err
python - <<PY
import csv, glob
f = glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("%-80s %6s %10s %8s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:12]:
    print("%-80s %6s %10.1f %8.2f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
