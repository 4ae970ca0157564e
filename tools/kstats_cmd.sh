#!/bin/bash
# per-kernel rocprofv3 averages of any command: tools/kstats_cmd.sh <name> <command ...>
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o p -- "$@" > /tmp/ks_$name.log 2>&1)
f=$(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
