#!/bin/bash
# SQ counters of the AMGe coarsening kernels (run on the GPU box from the repo root): where do the waves spend their cycles?
# usage: tools/pmc_amge.sh <outdir-under-gpurun_out> <tools/amge_time.py args...>
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/pmc -- python3 $REPO/tools/amge_time.py "$@" > $OUT/run.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "extension" if "coarsen_extension" in r["Kernel_Name"] else "traces" if "coarsen_traces" in r["Kernel_Name"] else None
        if k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    tot = {c: s for c, (s, n) in d.items()}
    n = max(v[1] for v in d.values())
    print("== %s kernel, %d dispatches (sums over all dispatches)" % (k, n))
    for c, s in sorted(tot.items()):
        print("   %-24s %.4e" % (c, s))
    wc = tot.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        print("   share of wave cycles: parked (s_waitcnt/barrier) %.1f%%  issue-stalled %.1f%%  issuing %.1f%%" %
              (100 * tot.get("SQ_WAIT_ANY", 0) / wc, 100 * tot.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * tot.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    if tot.get("SQ_LDS_IDX_ACTIVE"):
        print("   LDS bank-conflict cycles / LDS active cycles: %.1f%%" % (100 * tot.get("SQ_LDS_BANK_CONFLICT", 0) / tot["SQ_LDS_IDX_ACTIVE"]))
PY
