cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for C in ${SWEEP:-64 128 256 512}; do
  export EXO_GP_CHUNKS=$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3s_$C -o p -- python $R/tools/profile_gp.py ${CFG:-c3} > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/c3s_$C/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "celerite" in r["Name"]]
print("C=$C", {r["Name"].split("celerite_")[1][:14]: round(float(r["AverageNs"])/1e3) for r in rows[:6]}, "sum ms", round(sum(float(r["TotalDurationNs"]) for r in rows)/3e6,2))
PY
done
