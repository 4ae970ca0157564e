"""print the kernel timeline of the last bench step from a rocprofv3 kernel trace csv"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:80]}")
