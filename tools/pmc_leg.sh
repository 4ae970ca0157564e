# SQ counters of transit_runs_kernel for one leg of the C2 sweep: tools/pmc_leg.sh <variant|base> <leg> [draws]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
v=$1; leg=$2; D=${3:-1024}
if [ "$v" = base ]; then unset EXOPLANET_AMD_LIB; else export EXOPLANET_AMD_LIB=$R/tests/_build/variants/$v.so; fi
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH"; do
  d=$R/gpurun_out/pl_${v}_${leg}_$(echo $c | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $R/tools/run_leg.py $leg $D 8 > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections
res=collections.defaultdict(list); dur=[]
for f in glob.glob("$R/gpurun_out/pl_${v}_${leg}_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "transit_runs" in r["Kernel_Name"]: res[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$R/gpurun_out/pl_${v}_${leg}_*/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "transit_runs" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
m={k:sum(x)/len(x) for k,x in res.items()}
print("$v $leg D=$D kernel_us=%.1f"%(sum(dur)/len(dur)))
for k in sorted(m): print("  %-22s %.4g"%(k,m[k]))
if "SQ_WAVE_CYCLES" in m:
    wc=m["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_VMEM"):
        if k in m: print("  %-22s / WAVE_CYCLES = %.3f"%(k,m[k]/wc))
if "GRBM_GUI_ACTIVE" in m and dur: print("  effective clock GHz = %.3f"%(m["GRBM_GUI_ACTIVE"]/(sum(dur)/len(dur))/1e3))
PY
