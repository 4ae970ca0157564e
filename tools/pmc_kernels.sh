#!/bin/bash
# per-kernel averages of SQ / memory counters over one eager bench step -- each counter set in its own rocprofv3 run, with
# --kernel-trace only (what gpurun allows):   bash tools/pmc_kernels.sh <name> <kernel filter> <bench.py arguments...>
#   -> gpurun_out/r05/<name>_pmc.json and a table on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; filt=$2; shift 2
out=$R/gpurun_out/r05
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE")
i=0
for c in "${SETS[@]}"; do
  d=/tmp/pmc_${name}_$i; rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-extras --no-stats --no-graph > /dev/null 2>&1
  i=$((i+1))
done
python - "$name" "$filt" "$out" <<'PY'
import csv, glob, collections, json, sys
name, filt, out = sys.argv[1:4]
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_%s_*/**/*counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("/tmp/pmc_%s_0/**/*kernel_trace.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
table = {}
for k, cs in res.items():
    if filt and filt not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["us_under_pmc"] = sum(dur[k]) / max(len(dur[k]), 1)
    table[k[:100]] = m
json.dump(table, open("%s/%s_pmc.json" % (out, name), "w"), indent=1)
for k, m in sorted(table.items(), key=lambda kv: -kv[1]["us_under_pmc"])[:8]:
    print(k[:80])
    wc = m.get("SQ_WAVE_CYCLES")
    for c in sorted(m):
        extra = ""
        if wc and c.startswith("SQ_") and ("ACTIVE" in c or "WAIT" in c):
            extra = "  / WAVE_CYCLES = %.3f" % (m[c] / wc)
        print("   %-22s %.5g%s" % (c, m[c], extra))
PY
