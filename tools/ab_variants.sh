cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_$v -- python $R/tools/profile_scan.py >/dev/null 2>&1
  python - <<PY
import csv,glob,collections
for f in glob.glob("$R/gpurun_out/ab_$v/*/*kernel_trace.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "transit" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][28:60], r.get("Grid_Size_X") or r.get("Grid_Size"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,v in agg.items(): print("$v", k, len(v), round(sum(v[1:])/max(1,len(v)-1)/1000,1))
PY
done
