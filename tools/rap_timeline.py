#!/usr/bin/env python3
"""Timeline of the LAST Galerkin product in a rocprofv3 --kernel-trace of tools/rap_profile.py: usage rap_timeline.py <kt dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# products start with the transposition of P (transpose / col_hist kernels) -- take everything after the last large gap (> 2 ms: the host's print)
starts = [0]
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 1_000_000:
        starts.append(i)
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    n = r["Kernel_Name"].replace("void elag::k::(anonymous namespace)::", "").replace("elag::k::(anonymous namespace)::", "")[:70]
    print("%9.1f +%8.1f us grid %-9s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), n))
print("span %.1f us, kernels busy %.1f us, %d dispatches" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3, len(seg)))
