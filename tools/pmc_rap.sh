#!/bin/bash
# SQ counters of the SpGEMM kernels of the fine-level Galerkin product (run on the GPU box from the repo root): where do the waves of
# esc_wave_kernel spend their cycles?  Counters in a pass of their own (no trace domains beside --pmc).
# usage: tools/pmc_rap.sh <outdir-under-gpurun_out> [k]
OUT=$PWD/gpurun_out/$1; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/pmc -- python3 $REPO/tools/rap_profile.py ${2:-5} > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc2 -- python3 $REPO/tools/rap_profile.py ${2:-5} > $OUT/run2.log 2>&1
cd $REPO
grep "RAP ms" $OUT/run.log
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("pmc", "pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(esc_wave_kernel<\d+, \d+, (?:true|false))|(esc_hash_kernel<\d+, \d+, \d+, \d+, \d+)|(ell_hash_kernel<\d+, \d+, \d+, \d+, \d+, \d+)|(esc_hashu_kernel<\d+, \d+, \d+, \d+)|esc_compact_kernel|esc_classify_kernel|esc_kernel", r["Kernel_Name"])
            if m:
                a = acc[m.group(0)][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    d = acc[k]
    tot = {c: s for c, (s, n) in d.items()}
    n = max(v[1] for v in d.values())
    print("== %s>, %d dispatches (sums over all dispatches)" % (k, n))
    for c, s in sorted(tot.items()):
        print("   %-24s %.4e" % (c, s))
    wc = tot.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        print("   share of wave cycles: parked (s_waitcnt/barrier) %.1f%%  issue-stalled %.1f%%  issuing %.1f%%;  waves %.3e, cycles per wave %.0f" %
              (100 * tot.get("SQ_WAIT_ANY", 0) / wc, 100 * tot.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * tot.get("SQ_ACTIVE_INST_ANY", 0) / wc,
               tot.get("SQ_WAVES", 0), wc / max(tot.get("SQ_WAVES", 1), 1)))
    if tot.get("SQ_LDS_IDX_ACTIVE"):
        print("   LDS bank-conflict cycles / LDS active cycles: %.1f%%" % (100 * tot.get("SQ_LDS_BANK_CONFLICT", 0) / tot["SQ_LDS_IDX_ACTIVE"]))
    if tot.get("SQ_INSTS_VALU"):
        w = max(tot.get("SQ_WAVES", 1), 1)
        print("   per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM rd %.0f  wr %.0f" % tuple(tot.get(c, 0) / w for c in
              ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
PY
