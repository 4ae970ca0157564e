"""Turn a gpurun_out/<tag>/ directory (bench.json + rocprofv3 csv outputs) into the
small text summary that is committed under profiles/."""
import csv
import json
import os
import sys


def main(src, dst):
    out = []
    b = json.load(open(os.path.join(src, "bench.json")))
    out.append("# bench.py (un-profiled run)\n")
    out.append(json.dumps(b, indent=1))
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
    out.append("\n\n# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras\n")
    out.append("# (profiled passes clock ~5% lower than un-profiled ones: MI355X_MICROARCH.md, DVFS give-back)\n")
    out.append(f"{'kernel':<78} {'calls':>6} {'avg_us':>10} {'total_ms':>10} {'pct':>6}")
    for r in rows[:14]:
        out.append(f"{r['Name'][:78]:<78} {r['Calls']:>6} {float(r['AverageNs'])/1e3:>10.1f} "
                   f"{float(r['TotalDurationNs'])/1e6:>10.2f} {float(r['Percentage']):>6.2f}")
    for sub, name, cname in (("pmc_fetch", "fetch", "FETCH_SIZE"), ("pmc_write", "write", "WRITE_SIZE")):
        f = os.path.join(src, sub, f"{name}_counter_collection.csv")
        if not os.path.exists(f):
            continue
        agg = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and "transit" in r["Kernel_Name"]:
                agg.setdefault(r["Kernel_Name"][:70], []).append(float(r["Counter_Value"]))
        out.append(f"\n# rocprofv3 --pmc {cname} (own pass, eager launches); unit KiB per dispatch, mean over dispatches")
        for k, v in agg.items():
            out.append(f"{k:<72} n={len(v):<3} {cname}={sum(v)/len(v):.1f}")
    out.append("\n# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of a wide coalesced read stream;"
               "\n# traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch")
    open(dst, "w").write("\n".join(out) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
