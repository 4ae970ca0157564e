#!/bin/bash
# Round-2 measurement record for profiles/: rocprofv3 kernel stats of the bench command (eager launches so that every
# kernel is a dispatch), the HBM traffic of the sweep from two SEPARATE PMC passes (FETCH_SIZE, WRITE_SIZE; --pmc only
# with --kernel-trace, as gpurun requires), then the un-profiled bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r02_record}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stats --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- $CMD > $out/bench_traced.json 2> $out/bench_traced.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_ops -o p -- python $R/tools/profile_ops.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o write -- $CMD > /dev/null 2>&1
python - <<PY
import csv, glob, json, collections, re
out = "$out"
import hashlib
res = {"kernel_source_sha256": hashlib.sha256(open("$R/exoplanet_amd/csrc/exo_transit.hip", "rb").read()).hexdigest(),
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (each with --kernel-trace), "
                 "bench.py --steps 20 --warmup 3 --no-graph --no-extras --draws-per-gpu 1024; KiB per dispatch, mean",
       "draws": 1024, "n_cad": 150000, "kernels": {},
       "correction": "gfx950: FETCH_SIZE counts half the bytes of a wide coalesced read stream -> traffic = "
                     "(2*fetch + write) KiB (MI355X_MICROARCH.md, HBM section)"}
for sub, name, cname, key in (("pmc_fetch", "fetch", "FETCH_SIZE", "fetch_kib"), ("pmc_write", "write", "WRITE_SIZE", "write_kib")):
    fs = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and "transit" in r["Kernel_Name"]:
                k = re.search(r"transit_\w+(?:<[^>]*>)?", r["Kernel_Name"]).group(0)
                agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res["kernels"].setdefault(k, {})[key] = sum(v) / len(v)
        res["kernels"][k]["dispatches_" + name] = len(v)
for k in res["kernels"].values():
    k.setdefault("fetch_kib", 0.0); k.setdefault("write_kib", 0.0)
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
f = glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
lines = ["%-86s %6s %10s %8s" % ("kernel", "calls", "avg_us", "pct")]
for r in rows[:14]:
    lines.append("%-86s %6s %10.1f %8.2f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
f = glob.glob(f"{out}/trace_ops/**/*kernel_stats.csv", recursive=True)
if f:
    lines.append("# standalone ops at n = 1.5e8 (tools/profile_ops.py): exo_kepler_f64 32 B/elt, exo_quad_solution_vector_f64 48 / 96 B/elt")
    for r in csv.DictReader(open(f[0])):
        if "kepler_kernel" in r["Name"] or "quad_sv" in r["Name"]:
            lines.append("%-86s %6s %10.1f %8.2f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
open(f"{out}/kernel_stats.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
# the un-profiled bench line last, with this record's counters in place so that `roofline.traffic` is filled in
# (bench.py only quotes profiles/r02_pmc.json when it was taken on the exo_transit.hip it is running)
cp $out/pmc.json $R/profiles/r02_pmc.json
python $R/bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
