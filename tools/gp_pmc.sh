cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/gppmc_$c -o p -- python $R/tools/profile_gp.py c3 > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("$R/gpurun_out/gppmc_%s/**/*counter_collection.csv"%c,recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "celerite" in r["Kernel_Name"] and r["Counter_Name"]==c: agg[r["Kernel_Name"].split("celerite_")[1][:16]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(c,k,round(sum(v)/len(v)/1e6,3),"GiB-ish (KiB/1e6)")
PY
