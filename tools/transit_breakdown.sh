# dynamic instruction breakdown of transit_runs_kernel (C2, 1024 draws): counters of library variants with pieces of
# eval_sample stubbed out (tests/_build/variants/st_*.so, built with -DEXO_STUB_KEPLER / _SV / _LOCATE: wrong results,
# right instruction streams) -> instructions per solved cadence attributable to each piece
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
  CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-stats --no-graph"
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS; do
    rm -rf $R/gpurun_out/bd_${v}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/bd_${v}_$c -o p -- $CMD > /dev/null 2>&1
  done
  python - <<PY
import csv,glob,collections,re
res=collections.defaultdict(dict); dur=collections.defaultdict(list)
for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS"):
    fs=glob.glob("$R/gpurun_out/bd_${v}_%s/**/*counter_collection.csv"%c,recursive=True)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "transit_runs" in r["Kernel_Name"] and r["Counter_Name"]==c:
            agg["runs"].append(float(r["Counter_Value"]))
    for k,x in agg.items(): res[k][c]=sum(x)/len(x)
    for f in glob.glob("$R/gpurun_out/bd_${v}_%s/**/*kernel_trace.csv"%c,recursive=True):
        for r in csv.DictReader(open(f)):
            if "transit_runs" in r["Kernel_Name"]: dur["runs"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
n_solved=4.34e6
for k,x in res.items():
    print("$v", {a:("%.0f per 64 solved cadences"%(b/(n_solved/64))) for a,b in x.items()}, "kernel us (profiled) %.1f"%(sum(dur[k])/max(1,len(dur[k]))))
PY
done
