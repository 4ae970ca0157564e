R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r06w
cd /tmp && export TMPDIR=/tmp
for kk in rot2_sho sho4; do
rm -rf /tmp/prof_$kk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$kk -o p -- python $R/tools/wide_step.py $kk 128 3 > /dev/null 2>&1
f=$(find /tmp/prof_$kk -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r06w/${kk}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[1], 'total ms per step', tot/3/1e6)
for r in rows[:16]:
    print("%-100s %6s calls %10.1f us avg %6.2f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
