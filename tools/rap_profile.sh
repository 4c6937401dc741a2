#!/bin/bash
# usage: tools/rap_profile.sh <tag> [k]   -> kernel stats of tools/rap_profile.py under gpurun_out/<tag>
OUT=$PWD/gpurun_out/$1; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/rap_profile.py ${2:-5} > $OUT/log.txt 2>&1
cd $REPO
grep "RAP ms" $OUT/log.txt
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("  %-120s calls=%-5s total_ms=%-9.3f avg_us=%-9.1f" % (r["Name"][:120], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
