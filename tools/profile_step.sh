# kernel totals of one whole user-level step (C5 or C3 leg of bench.py), eager launches: where the non-GP time goes
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; CFG=${CFG:-c5}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/step_$CFG -o p -- python $R/tools/profile_c5_step.py $CFG > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/step_$CFG/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
gp=sum(float(r["TotalDurationNs"]) for r in rows if "celerite" in r["Name"])
tr=sum(float(r["TotalDurationNs"]) for r in rows if "transit" in r["Name"] or "pack" in r["Name"])
print("per step (5 steps, CFG=$CFG): total %.3f ms, celerite %.3f, transit+pack %.3f, other (torch) %.3f" % (tot/5e6, gp/5e6, tr/5e6, (tot-gp-tr)/5e6))
for r in rows[:14]:
    print("%-70s calls %4s total/step %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/5e3))
PY
