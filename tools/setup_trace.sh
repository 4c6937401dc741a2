#!/bin/bash
# HIP API and kernel statistics of hierarchy setup (run on the GPU box from the repo root)
# usage: tools/setup_trace.sh <outdir-under-gpurun_out> <setup_probe args...>
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/tr -- python3 $REPO/tools/setup_probe.py "$@" > $OUT/run.log 2>&1
cd $REPO
cat $OUT/run.log | tail -8
python3 - <<PY
import csv, glob
for pat, n in (("hip_api_stats", 14), ("kernel_stats", 25)):
    for f in sorted(glob.glob("$OUT/tr/**/*%s.csv" % pat, recursive=True)):
        print("==", pat)
        for r in list(csv.DictReader(open(f)))[:n]:
            print("  %-80s calls=%-7s total_us=%-10.1f avg_us=%-9.2f %s%%" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
