#!/bin/bash
# usage: tools/pmc_one.sh <variant> <counters...>   (stream-kernel variant A/B on cube456 x5)
V=$1; shift
REPO=$PWD; OUT=$PWD/gpurun_out/pmc_v$V; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ELAG_HIP_STREAM_VARIANT=$V rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $REPO/tools/spmv_tune.py 5 0 > $OUT/log.txt 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spmv_stream" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("variant $V", k, s / max(n, 1), n)
PY
