#!/bin/bash
# rocprofv3 passes over the fine-level SpMV (run on the GPU box from the repo root).
# usage: tools/profile_spmv.sh <refines> <lanes> <outdir-under-gpurun_out>
set -u
K=${1:-5}; L=${2:-8}; OUT=$PWD/gpurun_out/${3:-prof_spmv}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/tools/spmv_tune.py $K $L > $OUT/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "MemUnitStalled MemUnitBusy" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $REPO/tools/spmv_tune.py $K $L > $OUT/pmc_$N.log 2>&1
done
cd $REPO
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in sorted(glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)):
    print("== kernel stats", f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  ", r.get("Name","")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"))
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "spmv_" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in acc.items():
        print("pmc", k, "mean per dispatch", s / max(n, 1), "dispatches", n)
PY
