"""print the kernel timeline of one replayed bench step from a rocprofv3 kernel trace csv"""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last full step: find last 'transit_scan' and walk back to previous reduce end
idx = [i for i, r in enumerate(rows) if "transit_scan" in r["Kernel_Name"]]
lo = idx[-2]; hi = idx[-1]
# shift window to start at the first kernel after the previous step's last kernel
names = [r["Kernel_Name"] for r in rows]
start = lo
while start > 0 and "pack_vjp" not in names[start - 1] and start > idx[-3]: start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:start + (hi - lo)]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev_end = e
print("step span", (prev_end - t0) / 1e3, "us")
