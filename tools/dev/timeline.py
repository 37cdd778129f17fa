#!/usr/bin/env python3
"""dev: print the last N kernel dispatches of a rocprofv3 kernel-trace csv as a timeline (start offset, duration, gap)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
    prev_end = max(prev_end, e)
print("span us", (prev_end - t0) / 1e3)
