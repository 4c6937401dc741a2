#!/bin/bash
# which HIP calls does the FIRST hierarchy of a process pay that later ones do not?  (hip-trace of tools/setup_probe.py 5 2)
OUT=$PWD/gpurun_out/${1:-r04fb}; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/ht -- python3 $REPO/tools/setup_probe.py 5 2 > $OUT/log.txt 2>&1
cd $REPO
cat $OUT/log.txt | tail -3
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/ht/**/*hip_api_stats.csv", recursive=True)):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("  %-40s calls=%-7s total_ms=%-10.3f avg_us=%-10.1f max_us=%s" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r.get("MaxNs",0))/1e3))
PY
