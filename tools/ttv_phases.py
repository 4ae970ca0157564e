"""Per-phase kernel averages from a rocprofv3 kernel trace of tools/profile_ttv.py: the script runs
its timed legs one after the other (3 warm-up + 10 timed calls each), so a new leg starts at every
14th window-kernel launch.   python tools/ttv_phases.py <..._kernel_trace.csv>"""
import csv
import sys

CALLS_PER_LEG = 13   # 3 warm-up + 10 timed

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
legs, cur = [], {}
for r in rows:
    if "transit" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("::")[1].split("(")[0]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "window" in name and len(cur.get(name, [])) >= CALLS_PER_LEG:
        legs.append(cur)
        cur = {}
    cur.setdefault(name, []).append(us)
legs.append(cur)
for leg in legs:
    print({k.replace("transit_", "")[:40]: round(sum(v[3:]) / max(len(v[3:]), 1), 1)
           for k, v in leg.items() if "window" not in k and "reduce" not in k})
