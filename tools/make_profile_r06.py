#!/usr/bin/env python
"""gpurun_out/r06 (tools/profile_r06.sh) -> profiles/r06_counters.json + profiles/r06_bench_rocprof_summary.txt.

Per leg (c2 dense step, sparse / chi2 op legs, c3, c4 at 64 draws, c5 at 128 chains) and per kernel: rocprofv3 average
duration (kernel-trace), HBM traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half the bytes of a
wide coalesced read stream; MI355X_MICROARCH.md, HBM section), and the fp64-VALU view:
    insts_per_unit            SQ_INSTS_VALU per wave-instruction slot of one unit (a solved cadence, or a (draw, cadence))
    frac_of_fp64_issue_peak   SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz): every vector instruction
                              priced as one fp64 issue slot (a wave64 fp64 FMA occupies a SIMD for 4 cycles; 78.6 TFLOP/s)
    valu_busy_frac            SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / the same denominator: cycles the vector pipes
                              were actually executing (32-bit operations take 2 cycles, fp64 reciprocals 16)
    wave_state                SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY as fractions of SQ_WAVE_CYCLES
The 2.4 GHz in the denominators is the nominal clock: under fp64 load the chip runs below it, so the fractions are
lower bounds of the pipes' occupancy."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, CLK = 1024, 2.4e9
SOURCES = sorted(f for f in os.listdir(os.path.join(ROOT, "exoplanet_amd", "csrc")) if f.endswith((".hip", ".hpp")))   # every kernel source


def short(name):
    m = re.search(r"((?:transit|celerite|pack|sho|ttv|nuts|rv|orbit|kepler|quad)_\w+)(<[^(]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def trace(leg):
    """kernel -> (calls, average us) from the kernel trace of the --stats run"""
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{src}/{leg}_trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def counters(leg):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{src}/{leg}_pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in out.items()}


def leg_record(leg, units_of):
    tr, cn = trace(leg), counters(leg)
    kernels = {}
    for k, (calls, us) in sorted(tr.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        c = cn.get(k, {})
        rec = {"calls": calls, "rocprof_avg_us": us}
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rec["fetch_kib"], rec["write_kib"] = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
            rec["traffic_bytes"] = (2.0 * rec["fetch_kib"] + rec["write_kib"]) * 1024.0
            rec["traffic_GBps"] = rec["traffic_bytes"] / (us * 1e-6) / 1e9
        if "SQ_INSTS_VALU" in c:
            denom = N_SIMD * us * 1e-6 * CLK
            rec["valu_insts"], rec["salu_insts"], rec["waves"] = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU"), c.get("SQ_WAVES")
            rec["frac_of_fp64_issue_peak"] = c["SQ_INSTS_VALU"] * 4.0 / denom
            if "SQ_ACTIVE_INST_VALU" in c:
                rec["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / denom
            u = units_of(k)
            if u:
                rec["valu_insts_per_unit"] = c["SQ_INSTS_VALU"] * 64.0 / u
        if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_ANY" in c:
            wc = c["SQ_WAVE_CYCLES"]
            rec["wave_state"] = {"active": c["SQ_ACTIVE_INST_ANY"] / wc, "wait_inst": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                 "wait_any": c.get("SQ_WAIT_ANY", 0) / wc}
        kernels[k] = rec
    return kernels


def waves_per_simd(kernel):
    """from the register / LDS budgets of the build (tools/kernel_stats.py): what the launch bounds ask for"""
    if kernel.startswith("transit_runs_kernel"):
        return 2 if ("true, false, true" in kernel or "true, true, true" in kernel or kernel.rstrip(">").endswith("true")) else 3
    return None


h = hashlib.sha256()
for f in SOURCES:
    h.update(open(os.path.join(ROOT, "exoplanet_amd", "csrc", f), "rb").read())
rec = {"kernel_sources": SOURCES, "kernel_sources_sha256": h.hexdigest(),
       "how": "tools/profile_r06.sh: rocprofv3 --kernel-trace --stats for the durations; --pmc passes one counter set per "
              "run; gfx950 FETCH_SIZE correction x 2; denominators at the nominal 2.4 GHz (lower bounds)"}
N_C2, D_C2 = 150_000, 1024
solved = None
try:
    solved = json.loads(open(f"{src}/bench_pre.json").read().strip().split("\n")[-1])["roofline"]["active_cadences_per_launch"]
except Exception:
    pass
solved = solved or 4.34e6
lines = []
for leg, D, N, J in (("c2", 1024, 150_000, 0), ("sparse", 1024, 150_000, 0), ("chi2", 1024, 150_000, 0), ("c3", 1024, 150_000, 2),
                     ("c4", 64, 200_000, 0), ("c5", 128, 65_000, 6), ("c5b", 128, 65_000, 6),
                     ("j8", 128, 65_000, 8), ("j10", 128, 65_000, 10),
                     ("kepler", 1, 150_000_000, 0), ("quadsv", 1, 150_000_000, 0)):
    def units_of(k, D=D, N=N, leg=leg):
        if k.startswith("transit_runs_kernel") and leg in ("c2", "sparse", "chi2"):
            return solved
        if k.startswith("celerite_chunk") or k.startswith("celerite_elem"):
            return D * N
        if leg in ("kepler", "quadsv") and (k.startswith("kepler_") or k.startswith("quad_sv")):
            return N       # (a lane of the pair kernels holds two elements: instructions per ELEMENT = insts x 64 / N all the same)
        return None
    ks = leg_record(leg, units_of)
    if not ks:
        continue
    step_kernels = {k: v for k, v in ks.items() if v["calls"] >= (2 if leg in ("kepler", "quadsv") else 3)}
    dom = max(step_kernels.items(), key=lambda kv: kv[1]["calls"] * kv[1]["rocprof_avg_us"])
    # per step: a kernel launched k times per step (the levels of the scan trees) counts k times
    total_tr = sum(v.get("traffic_bytes", 0.0) * (v["calls"] / dom[1]["calls"]) for v in step_kernels.values())
    entry = {"draws": D, "n_cadences": N, "source": f"profiles/r06_counters.json [{leg}] (tools/profile_r06.sh)", "kernels": ks,
             "dominant_kernel": {"name": dom[0], **{k: dom[1].get(k) for k in ("rocprof_avg_us", "traffic_bytes", "traffic_GBps")}}}
    if leg in ("kepler", "quadsv"):
        entry["algorithmic_bytes"] = {"kepler_pair_kernel": 32.0 * N, "kepler_kernel": 32.0 * N, "quad_sv_kernel<false>": 40.0 * N, "quad_sv_kernel<true>": 88.0 * N,
                                      "quad_sv_pair_kernel<false>": 40.0 * N, "quad_sv_pair_kernel<true>": 88.0 * N}
        for k, v in ks.items():
            ab = entry["algorithmic_bytes"].get(k)
            if ab:
                v["algorithmic_GBps"] = ab / (v["rocprof_avg_us"] * 1e-6) / 1e9
                v["algorithmic_frac_of_hbm_peak"] = v["algorithmic_GBps"] / 8000.0
    if leg in ("c2", "c4"):
        entry["traffic_bytes_per_sweep"] = sum(v.get("traffic_bytes", 0.0) for k, v in step_kernels.items() if k.startswith("transit_"))
    else:
        entry["traffic_bytes_per_step"] = total_tr
    valu = {}
    for k, v in step_kernels.items():
        if "frac_of_fp64_issue_peak" in v and v["calls"] * v["rocprof_avg_us"] > 0.1 * dom[1]["calls"] * dom[1]["rocprof_avg_us"]:
            valu[k] = {"insts_per_solved_cadence" if k.startswith("transit_runs") else "insts_per_draw_cadence": v.get("valu_insts_per_unit"),
                       "waves_per_simd": waves_per_simd(k), "frac_of_fp64_issue_peak": v["frac_of_fp64_issue_peak"],
                       "valu_busy_frac": v.get("valu_busy_frac"), "rocprof_avg_us": v["rocprof_avg_us"], "wave_state": v.get("wave_state")}
    entry["valu"] = valu
    rec[leg] = entry
    lines.append(f"# {leg}: D = {D}, N = {N}" + (f", J = {J}" if J else "") + (" -- 2 chains at a conditioning score of 1e6 (robust route)" if leg == "c5b" else ""))
    lines.append("%-64s %6s %10s %12s %10s %8s %8s" % ("kernel", "calls", "avg_us", "traffic_MB", "GB/s", "valu/pk", "busy"))
    for k, v in list(ks.items())[:16]:
        lines.append("%-64s %6d %10.1f %12s %10s %8s %8s" % (
            k[:64], v["calls"], v["rocprof_avg_us"], "%.1f" % (v["traffic_bytes"] / 1e6) if "traffic_bytes" in v else "-",
            "%.0f" % v["traffic_GBps"] if "traffic_GBps" in v else "-",
            "%.2f" % v["frac_of_fp64_issue_peak"] if "frac_of_fp64_issue_peak" in v else "-",
            "%.2f" % v["valu_busy_frac"] if "valu_busy_frac" in v else "-"))
    lines.append("")
# the sparse / chi2 op legs belong to the C2 record's VALU block
for leg in ("sparse", "chi2"):
    if leg in rec and "c2" in rec:
        for k, v in rec[leg]["valu"].items():
            rec["c2"]["valu"][f"{k} [{leg} leg]"] = v
os.makedirs(dst, exist_ok=True)
json.dump(rec, open(os.path.join(dst, "r06_counters.json"), "w"), indent=1)
open(os.path.join(dst, "r06_bench_rocprof_summary.txt"), "w").write(
    "rocprofv3 record of round 6 (tools/profile_r06.sh; eager launches so that every kernel is a dispatch).  avg_us: kernel-trace\n"
    "average; traffic: (2 x FETCH_SIZE + WRITE_SIZE) per dispatch; valu/pk: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x avg x 2.4 GHz);\n"
    "busy: SQ_ACTIVE_INST_VALU x 4 / the same.  Full numbers: r06_counters.json.\n\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
