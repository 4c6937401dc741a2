#!/usr/bin/env python3
"""One V-cycle out of a rocprofv3 --kernel-trace of bench.py (the second-to-last cycle of the timed loop): usage cycle_timeline.py <kt dir> [first-kernel-substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else "gather_scale"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void elag::k::(anonymous namespace)::", "").replace("elag::k::(anonymous namespace)::", "")[:60]
    print("%8.1f +%7.1f us  grid %-9s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), n))
