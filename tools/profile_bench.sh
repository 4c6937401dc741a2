#!/bin/bash
# rocprofv3 kernel trace of bench.py (run on the GPU box from the repo root)
# usage: tools/profile_bench.sh <outdir-under-gpurun_out> <bench args...>
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (ELAG_HIP_GMRES_GRAPH=0: the kernel trace faults inside hipGraphLaunch on the GMRES step graphs of the other_configs child -- seen in this round)
ELAG_HIP_GMRES_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py "$@" > $OUT/bench.log 2>&1
cd $REPO
tail -1 $OUT/bench.log | cut -c1-600
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)):
    print("== kernel stats", f)
    for r in list(csv.DictReader(open(f)))[:25]:
        print("  %-95s calls=%-7s total_ns=%-12s avg_ns=%-10s %s%%" % (r["Name"][:95], r["Calls"], r["TotalDurationNs"], r["AverageNs"][:9], r["Percentage"]))
PY
