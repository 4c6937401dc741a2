#!/bin/bash
# kernel timeline of the last hierarchy build of tools/setup_probe.py <k> <n>: busy time, gaps, the longest of both
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/tools/${PROBE:-setup_probe.py} "$@" > $OUT/run.log 2>&1
cd $REPO
tail -n 1 $OUT/run.log | cut -c1-250
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True))[0]
k = list(csv.DictReader(open(f)))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(k) if "rap_fused" in r["Kernel_Name"] or "ell_hash_kernel<8" in r["Kernel_Name"]]
# the last build starts at the last product kernel that follows a gap of > 200 us
starts = [i for i in idx if i > 3 and int(k[i - 3]["Start_Timestamp"]) - int(k[i - 4]["End_Timestamp"]) > 200000]
i0 = (starts[-1] - 3) if starts else 0
seg = k[i0:]
t0 = int(seg[0]["Start_Timestamp"]); prev = None; busy = 0; rows = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("elag::k::(anonymous namespace)::", "").replace("elag::(anonymous namespace)::", "").replace("void ", "")[:48]
    rows.append(((s - t0) / 1e3, (e - s) / 1e3, gap, name)); prev = e
span = (prev - t0) / 1e3
print("last build: span %.1f us, kernels busy %.1f us, %d dispatches, gaps > 8 us: %d totalling %.1f us" % (span, busy / 1e3, len(rows), sum(1 for r in rows if r[2] > 8), sum(r[2] for r in rows if r[2] > 8)))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for t, d, g, n in rows: agg[n][0] += 1; agg[n][1] += d
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]: print("  %-50s x%-3d %.1f us" % (n, c, d))
open("$OUT/timeline.txt", "w").write("\n".join("%8.1f +%6.1f gap %5.1f %s" % r for r in rows))
PY
