#!/bin/bash
# kernel trace of tools/r05_gmres_chain_probe.py: the chain kernel's duration against the step number (run on the GPU box from the repo root)
OUT=$PWD/gpurun_out/${1:-chain_prof}; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ELAG_HIP_GMRES_GRAPH=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/tools/r05_gmres_chain_probe.py ${2:-706000} ${3:-50} > $OUT/probe.log 2>&1
cd $REPO
cat $OUT/probe.log | tail -5
python3 - <<PY
import csv, glob, numpy as np
for f in sorted(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    for key in ("gmres_chain_kernel<6", "gmres_chain_kernel<12", "gmres_orth_fused_kernel"):
        d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if key in r["Kernel_Name"]]
        if not d: continue
        d.sort()
        dur = np.array([x[1] for x in d], float) / 1e3
        if "chain" in key:
            m = ${3:-50}
            cyc = dur[-m:]                       # the last solve's cycle: steps 0 .. m-1
            i = np.arange(m)
            slope, icpt = np.polyfit(i[5:], cyc[5:], 1)
            print("%s: %d dispatches; last cycle: step 0 %.2f us, step 10 %.2f, step 25 %.2f, step %d %.2f; %.3f us per link, %.2f us fixed" % (key, len(d), cyc[0], cyc[10], cyc[25], m - 1, cyc[-1], slope, icpt))
        else:
            print("%s: %d dispatches, mean %.2f us, median %.2f" % (key, len(d), dur.mean(), np.median(dur)))
PY
