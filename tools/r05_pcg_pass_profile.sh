#!/bin/bash
# kernel statistics of tools/r05_pcg_pass_probe.py (PCG passes of the headline-size solve, three launches / one): run on the GPU box from the repo root
OUT=$PWD/gpurun_out/${1:-pcgprof}; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/r05_pcg_pass_probe.py ${2:-3} ${3:-256} > $OUT/probe.log 2>&1
cd $REPO
grep -a "passes\|iterate" $OUT/probe.log
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)):
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-100s calls=%-7s avg_us=%-8.2f %s%%" % (r["Name"].replace("elag::k::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $OUT/kt
