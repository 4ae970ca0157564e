# A/B of library variants on the small-batch legs (C4 at 64 draws op level; C2 step at 64 / 128 / 256 draws)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
  for leg in "c4_64 tools/profile_c4.py 64" "c2_64 bench.py --steps 20 --warmup 3 --no-graph --no-extras --no-stats --draws-per-gpu 64" "c2_128 bench.py --steps 20 --warmup 3 --no-graph --no-extras --no-stats --draws-per-gpu 128" "c2_256 bench.py --steps 20 --warmup 3 --no-graph --no-extras --no-stats --draws-per-gpu 256"; do
    set -- $leg; tag=$1; shift
    rm -rf $R/gpurun_out/small_${v}_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small_${v}_$tag -o p -- python $R/$@ > /dev/null 2>&1
    python - <<PY
import csv,glob,re
f=glob.glob("$R/gpurun_out/small_${v}_$tag/**/*kernel_stats.csv",recursive=True)[0]
out={}
for r in csv.DictReader(open(f)):
    m=re.search(r'(transit_\w+)(<[^>]*>)?',r["Name"])
    if m and int(r["Calls"])>2: out[m.group(0)[8:30]]=round(float(r["AverageNs"])/1e3,1)
print("$v $tag", out, "sum", round(sum(out.values()),1))
PY
  done
done
