#!/bin/bash
# rocprofv3 kernel stats of the GP legs of C3 and C5 (tools/profile_gp.py), time-parallel and sequential
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${GP_PROF_TAG:-gp_prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c5; do
  for mode in ${GP_PROF_MODES:-chunked sequential}; do
    if [ $mode = sequential ]; then export EXO_GP_CHUNKS=1; else unset EXO_GP_CHUNKS; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${cfg}_$mode -o p -- python $R/tools/profile_gp.py $cfg > /dev/null 2>&1
  done
done
unset EXO_GP_CHUNKS
python - <<PY
import csv, glob
out = []
for cfg in ("c3", "c5"):
    for mode in ("chunked", "sequential"):
        f = glob.glob("$out/%s_%s/**/*kernel_stats.csv" % (cfg, mode), recursive=True)
        if not f: continue
        rows = [r for r in csv.DictReader(open(f[0])) if "celerite" in r["Name"]]
        out.append("# %s GP leg, %s path: rocprofv3 --kernel-trace --stats -- python tools/profile_gp.py %s (3 value+grad steps)" % (cfg.upper(), mode, cfg))
        tot = 0.0
        for r in rows:
            out.append("%-72s calls %3s  avg %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
            tot += float(r["TotalDurationNs"]) / 3e6
        out.append("sum of celerite kernels per step: %.2f ms\n" % tot)
open("$out/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
