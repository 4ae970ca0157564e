"""profiles/r02_* from a tools/profile_r02.sh record (gpurun_out/<tag>) and a tools/profile_gp.sh record:
    python tools/make_profile_summary.py gpurun_out/r02_final gpurun_out/r02_final_gp"""
import json, os, shutil, sys

rec, gp = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.loads(open(os.path.join(rec, "bench.json")).read().strip().splitlines()[-1])
pmc = json.load(open(os.path.join(rec, "pmc.json")))
stats = open(os.path.join(rec, "kernel_stats.txt")).read()
sweep = {k: v for k, v in pmc["kernels"].items() if v.get("dispatches_fetch", 0) > 5}
tot = sum(2.0 * v["fetch_kib"] + v["write_kib"] for v in sweep.values()) * 1024.0
req = bench["roofline"]["algorithmic_bytes_per_launch"]
kms = bench["roofline"]["kernel_ms"]
lines = ["# Round 2, C2 sweep on one MI355X (tools/profile_r02.sh).  1) un-profiled bench.py line",
         json.dumps(bench, indent=1), "",
         "# 2) rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-stats --no-graph",
         "#    (eager launches: every kernel of a step is a dispatch; profiled passes clock a few % lower than un-profiled ones)",
         stats.rstrip(), "",
         "# 3) HBM traffic, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, KiB per dispatch (mean):"]
for k, v in sweep.items():
    lines.append("#   %-44s FETCH_SIZE %12.1f  WRITE_SIZE %12.1f" % (k, v["fetch_kib"], v["write_kib"]))
lines += ["# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of a wide coalesced read stream;",
          "# traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 = %.3f GB per sweep of 1024 draws x 150 000 cadences" % (tot / 1e9),
          "# bytes the design must move (bench.py roofline.algorithmic_bytes_per_launch) = %.3f GB: traffic / required = %.3f"
          % (req / 1e9, tot / req),
          "# sweep kernels (hipEvents, un-profiled) %.4f ms -> %.0f GB/s = %.3f of the 8 TB/s peak"
          % (kms, tot / (kms * 1e-3) / 1e9, tot / (kms * 1e-3) / 8e12)]
open(os.path.join(R, "profiles", "r02_bench_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(pmc, open(os.path.join(R, "profiles", "r02_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(gp, "summary.txt"), os.path.join(R, "profiles", "r02_gp_kernels_rocprof_summary.txt"))
print("traffic %.3f GB, required %.3f GB, kernel %.4f ms" % (tot / 1e9, req / 1e9, kms))
