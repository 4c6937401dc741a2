#!/bin/bash
# does rocprofv3's kernel trace survive the GMRES step graphs?  (the round's first evidence run lost its kernel statistics to a SIGSEGV inside run_gmres)
OUT=$PWD/gpurun_out/gmtr; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 0 1; do
  ELAG_HIP_GMRES_GRAPH=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$g -- python3 $REPO/bench.py --form darcy --solve --no-cpu-baseline --no-cpu-solve --steps 5 --warmup 2 > $OUT/g$g.log 2>&1
  echo "graph=$g rc=$?"; grep -c "SIGSEGV" $OUT/g$g.log; tail -n 1 $OUT/g$g.log | cut -c1-300
done
