# dynamic instruction counts of the light-curve sweep kernels (C2, 1024 draws): SQ_INSTS_VALU / SALU / LDS / waves per dispatch
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-stats --no-graph"
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/trinst_$c -o p -- $CMD > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,re
res=collections.defaultdict(dict)
for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVES","SQ_INSTS_VMEM_WR","SQ_INSTS_VMEM_RD","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VALU"):
    fs=glob.glob("$R/gpurun_out/trinst_%s/**/*counter_collection.csv"%c,recursive=True)
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "transit_runs" in r["Kernel_Name"] and r["Counter_Name"]==c:
            agg[re.search(r"transit_\w+(?:<[^>]*>)?",r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
    for k,v in agg.items(): res[k][c]=sum(v)/len(v)
for k,v in res.items():
    w=v.get("SQ_WAVES",1)
    print(k, {a:("%.3g"%b) for a,b in v.items()}, "per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_WR %.0f" % (v.get("SQ_INSTS_VALU",0)/w, v.get("SQ_INSTS_SALU",0)/w, v.get("SQ_INSTS_LDS",0)/w, v.get("SQ_INSTS_VMEM_WR",0)/w))
PY
