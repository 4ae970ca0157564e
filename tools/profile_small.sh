# kernel averages of the small-batch legs: C4 shape at 64 draws (op level), C2 step at 64 / 128 draws
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() {  # tag, command...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/small_$tag -o p -- "$@" > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/small_$tag/**/*kernel_stats.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if ("transit" in r["Name"] or "pack" in r["Name"])]
print("$tag", {r["Name"].split("::")[-1][:22]: (int(r["Calls"]), round(float(r["AverageNs"])/1e3,1)) for r in rows})
PY
}
run c4_64 python $R/tools/profile_c4.py 64
run c2_64 python $R/bench.py --steps 20 --warmup 3 --no-graph --no-extras --no-stats --draws-per-gpu 64
run c2_128 python $R/bench.py --steps 20 --warmup 3 --no-graph --no-extras --no-stats --draws-per-gpu 128
