#!/bin/bash
# one GMRES step of the Darcy x3 solve, kernel by kernel (plain launches: the kernel trace does not survive the step graphs) -- run on the GPU box from the repo root
OUT=$PWD/gpurun_out/${1:-dtl}; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ELAG_HIP_GMRES_GRAPH=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --form ${2:-darcy} --solve --no-cpu-baseline --no-cpu-solve --no-roofline --steps 5 --warmup 2 > $OUT/bench.log 2>&1
echo "rc $?"
cd $REPO
tail -n 1 $OUT/bench.log | cut -c1-300
python3 tools/cycle_timeline.py $OUT/kt gmres_chain_kernel > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
