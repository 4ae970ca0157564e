#!/usr/bin/env python
"""kernel trace of the forced 1-rank RCCL run of bench.py (EXO_BENCH_FORCE_DIST=1) -> does the step's collective run
WHILE the next step's kernels run?  usage: python tools/trace_overlap.py <rocprofv3 output dir> <summary.txt>"""
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower()  # noqa: E731
coll = [r for r in rows if is_coll(r[2])]
runs = [r for r in rows if "transit_runs_kernel" in r[2]]
lines = [f"{len(rows)} dispatches, {len(coll)} RCCL kernels, {len(runs)} transit_runs_kernel"]
if coll and runs:
    n_over, tot = 0, 0
    ex = []
    for c in coll[len(coll) // 2:]:          # (the timed half of the run)
        tot += 1
        ov = [(min(c[1], r[1]) - max(c[0], r[0])) for r in runs if r[0] < c[1] and r[1] > c[0]]
        if ov:
            n_over += 1
            if len(ex) < 6:
                r = next(r for r in runs if r[0] < c[1] and r[1] > c[0])
                ex.append(f"  RCCL [{c[0]}, {c[1]}] ({(c[1] - c[0]) / 1e3:.1f} us, queue {c[4]}) inside transit_runs_kernel "
                          f"[{r[0]}, {r[1]}] ({(r[1] - r[0]) / 1e3:.1f} us, queue {r[4]})")
    lines.append(f"collectives of the second half of the run that execute while a transit_runs_kernel executes: {n_over} of {tot}")
    lines += ex
    lines.append("RCCL kernel names: " + "; ".join(sorted({c[2][:80] for c in coll})))
    lines.append("mean RCCL kernel duration: %.1f us" % (sum(c[1] - c[0] for c in coll) / len(coll) / 1e3))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
