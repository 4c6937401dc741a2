#!/bin/bash
# round 4, end of round: the evidence behind DESIGN.md's numbers, from the final tree, in one GPU call
OUT=$PWD/gpurun_out/r04final; mkdir -p $OUT
echo "[1] default bench (the driver's command)"; timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc $?"
echo "[2] kernel stats of the same"; bash tools/profile_bench.sh r04final/prof --no-traffic --no-cpu-baseline --no-gs > $OUT/bench_default_kernel_stats.txt 2>&1; tail -3 $OUT/bench_default_kernel_stats.txt | cut -c1-160
echo "[3] roofline kernel: rocprof mean + PMC traffic"; bash tools/roofline_profile.sh r04final/roof auto > $OUT/roofline.txt 2>&1; tail -6 $OUT/roofline.txt | cut -c1-200
echo "[4] RAP"; bash tools/rap_profile.sh r04final/rap 5 > $OUT/rap_stats.txt 2>&1; python3 tools/rap_timeline.py $OUT/rap/kt > $OUT/rap_timeline.txt 2>&1; tail -1 $OUT/rap_timeline.txt; grep "RAP ms" $OUT/rap_stats.txt
echo "[5] setup probes"; python tools/setup_probe.py 3 4 > $OUT/setup_k3.txt 2>&1; python tools/setup_probe.py 5 4 > $OUT/setup_k5.txt 2>&1; tail -2 $OUT/setup_k5.txt | cut -c1-200
ELAG_HIP_LU_TIMING=1 python tools/r04_darcy_setup_probe.py 3 4 > $OUT/darcy_setup.txt 2>&1; tail -3 $OUT/darcy_setup.txt
echo "[6] H(div) x3 kernel stats"; bash tools/profile_bench.sh r04final/hdiv --form 2 --no-cpu-baseline --no-roofline --no-gs --no-other-configs > $OUT/hdiv_kernel_stats.txt 2>&1; tail -2 $OUT/hdiv_kernel_stats.txt | cut -c1-160
echo "[7] the N-rank code path on one rank"; timeout -k 10 400 python bench.py --force-dist --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_multirank_path_1rank.json 2> $OUT/bench_multirank_path_1rank.err; echo "rc $?"
echo "[8] config 4 with its solve"; timeout -k 10 300 python bench.py --form darcy --solve --no-cpu-baseline --no-cpu-solve --steps 20 --warmup 5 > $OUT/bench_darcy_solve.json 2> $OUT/bench_darcy_solve.err; echo "rc $?"
