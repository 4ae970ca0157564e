# dynamic instruction counts of the GP kernels (C3 or C5 leg): SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES per dispatch
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; CFG=${CFG:-c3}
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/gpinst_$c -o p -- python $R/tools/profile_gp.py $CFG > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections
res=collections.defaultdict(dict)
for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_WAVES","SQ_INSTS_VMEM_RD","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_BUSY_CYCLES"):
    fs=glob.glob("$R/gpurun_out/gpinst_%s/**/*counter_collection.csv"%c,recursive=True)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "celerite" in r["Kernel_Name"] and r["Counter_Name"]==c: agg[r["Kernel_Name"].split("celerite_")[1][:22]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): res[k][c]=sum(v)/len(v)
for k,v in res.items():
    if v.get("SQ_WAVES",0)>100: print(k, {a:("%.3g"%b) for a,b in v.items()}, "VALU/wave %.0f SALU/wave %.0f" % (v.get("SQ_INSTS_VALU",0)/v["SQ_WAVES"], v.get("SQ_INSTS_SALU",0)/v["SQ_WAVES"]))
PY
