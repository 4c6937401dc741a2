#!/bin/bash
# Kernel sequence of the V-cycle loop of a short bench run: rocprofv3 kernel trace, then <count> consecutive dispatches from the middle of the
# timed loop with their durations and the gaps between them (run on the GPU box from the repo root; ELAG_HIP_GRAPH_MODE=graph | eager picks
# how the cycle is issued).
# usage: tools/cycle_trace.sh <outdir-under-gpurun_out> <count> <bench.py args...>
OUT=$PWD/gpurun_out/$1; shift
CNT=$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
python3 - "$OUT" "$CNT" <<'PY'
import csv, glob, sys
out, cnt = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the V-cycle loop: the longest stretch of dispatches without a copy / fill kernel in it (the bench's timed loop runs on device vectors);
# print <count> dispatches from its middle
best, start = (0, 0), 0
for i, r in enumerate(rows + [(0, 0, "copyBuffer")]):
    if "copyBuffer" in r[2] or "fillBuffer" in r[2]:
        if i - start > best[1] - best[0]:
            best = (start, i)
        start = i + 1
mid = (best[0] + best[1]) // 2
tail = rows[mid:mid + cnt]
prev = None
for s, e, name in tail:
    short = name.replace("(anonymous namespace)::", "").replace("elag::k::", "").replace("elag::", "").replace("void ", "").split("(")[0][:70]
    print("%8.2f us  gap %6.2f us  %s" % ((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, short))
    prev = e
print("span of these %d dispatches: %.2f us" % (cnt, (tail[-1][1] - tail[0][0]) / 1e3))
PY
