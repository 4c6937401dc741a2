#!/bin/bash
# rocprofv3 kernel trace of the AMGe coarsening (run on the GPU box from the repo root)
# usage: tools/profile_amge.sh <outdir-under-gpurun_out> <tools/amge_time.py args...>     e.g.  amge_tet3 tet 3
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/amge_time.py "$@" > $OUT/run.log 2>&1
cd $REPO
tail -8 $OUT/run.log
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)):
    print("== kernel stats", f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-70s calls=%-5s total_ns=%-12s avg_ns=%-10s %s%%" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"][:9], r["Percentage"]))
PY
