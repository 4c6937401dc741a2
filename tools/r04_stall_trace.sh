#!/bin/bash
# where does the one 30-80 ms stall of a Darcy solve shortly after setup sit?  kernel trace (plain launches), longest kernels and longest gaps
OUT=$PWD/gpurun_out/stall; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ELAG_HIP_GMRES_GRAPH=0 ELAG_HIP_KRYLOV_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --form darcy --solve --no-cpu-baseline --no-cpu-solve --steps 20 --warmup 5 > $OUT/run.log 2>&1
cd $REPO
grep "krylov mult" $OUT/run.log | tr "\n" " "; echo
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True))[0]
k = list(csv.DictReader(open(f)))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(k[0]["Start_Timestamp"])
longk = sorted(k, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:8]
print("longest kernels:")
for r in longk:
    print("  at %.1f ms  %.2f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:90]))
gaps = []
for a, b in zip(k, k[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), a, b))
gaps.sort(key=lambda g: g[0], reverse=True)
print("longest gaps:")
for g, a, b in gaps[:10]:
    print("  at %.1f ms  gap %.2f ms  after %s  before %s" % ((int(a["End_Timestamp"]) - t0) / 1e6, g / 1e6, a["Kernel_Name"][-60:], b["Kernel_Name"][-60:]))
PY
