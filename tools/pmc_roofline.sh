#!/bin/bash
# SQ counters of the roofline kernel (bench.py --roofline-only, the library's own choice of form and window): where the waves of the
# fine-level SpMV spend their cycles.  Run on the GPU box from the repo root; counters in their own pass (no trace domains beside them).
# usage: tools/pmc_roofline.sh <outdir-under-gpurun_out> [bench args...]
OUT=$PWD/gpurun_out/$1; shift
REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py --roofline-only "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
bench = json.loads([l for l in open(out + "/run.log") if l.startswith("{")][-1])["roofline"]
kern = bench["kernel"]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and "probe" not in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
n = max(v[1] for v in acc.values())
tot = {c: s / n for c, (s, n_) in acc.items()}
print("# SQ counters of %s (%s entries per window), level-0 operator of the cube456 x5 hierarchy (n=%d, nnz=%d): means per launch over %d launches"
      % (kern, bench.get("window"), bench["n"], bench["nnz"], n))
print("# HIP events in the same run: %.2f us per launch, %.1f GB/s algorithmic" % (bench["ms_per_launch"] * 1e3, bench["achieved_GBs"]))
for c, v in sorted(tot.items()):
    print("   %-24s %.4e" % (c, v))
wc = tot.get("SQ_WAVE_CYCLES", 0.0)
if wc:
    print("   share of wave cycles: waiting (s_waitcnt / barrier) %.1f %%   issue-stalled %.1f %%   issuing %.1f %%" %
          (100 * tot.get("SQ_WAIT_ANY", 0) / wc, 100 * tot.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * tot.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    print("   wave cycles per wave: %.0f   (waves per launch %.0f)" % (wc / max(tot.get("SQ_WAVES", 1), 1), tot.get("SQ_WAVES", 0)))
if tot.get("SQ_LDS_IDX_ACTIVE"):
    print("   LDS bank-conflict cycles / LDS active cycles: %.1f %%" % (100 * tot.get("SQ_LDS_BANK_CONFLICT", 0) / tot["SQ_LDS_IDX_ACTIVE"]))
PY
