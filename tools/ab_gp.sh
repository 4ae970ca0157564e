# A/B of library variants (tests/_build/variants/<name>.so, same ABI) on the C5 / C3 GP legs: kernel averages
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abgp_$v -o p -- python $R/tools/profile_gp.py ${CFG:-c5} > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/abgp_$v/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "celerite" in r["Name"]]
tree = sum(float(r["TotalDurationNs"]) for r in rows if "tree_kernel" in r["Name"]) / 3e3
print("$v", {r["Name"].split("celerite_")[1][:14]: round(float(r["AverageNs"])/1e3) for r in rows[:6] if "tree_kernel" not in r["Name"]}, "trees us", round(tree), "sum ms", round(sum(float(r["TotalDurationNs"]) for r in rows)/3e6,2))
PY
done
