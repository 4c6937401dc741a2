#!/bin/bash
# round 5: the evidence behind DESIGN.md's numbers, from one tree, in one GPU call (profiles/r05_* are copies of what this writes)
OUT=$PWD/gpurun_out/r05final; mkdir -p $OUT
echo "[1] default bench (the driver's command)"; timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc $?"
echo "[2] kernel stats of the same"; bash tools/profile_bench.sh r05final/prof --no-traffic --no-cpu-baseline --no-gs > $OUT/bench_default_kernel_stats.txt 2>&1; tail -3 $OUT/bench_default_kernel_stats.txt | cut -c1-160
# (the Gauss-Seidel leg apart and as plain launches: rocprofv3 --kernel-trace faults inside hipGraphLaunch on this stack, as it did on the GMRES step graphs in round 4)
echo "[2b] kernel stats of the Gauss-Seidel cycles"; bash tools/profile_bench.sh r05final/gs --no-roofline --no-cpu-baseline --no-other-configs --no-graph > $OUT/gs_kernel_stats.txt 2>&1; grep -a "dataflow\|sweep_levels\|sweep_chain" $OUT/gs_kernel_stats.txt | cut -c1-200
echo "[3] roofline kernel: rocprof mean + PMC traffic"; bash tools/roofline_profile.sh r05final/roof auto > $OUT/roofline.txt 2>&1; tail -6 $OUT/roofline.txt | cut -c1-200
echo "[4] RAP"; bash tools/rap_profile.sh r05final/rap 5 > $OUT/rap_stats.txt 2>&1; grep "RAP ms" $OUT/rap_stats.txt
echo "[5] the Gauss-Seidel dataflow sweep: level operators of cube456 x3, the pure chain, the cycle's kernel stats"
python tools/r05_dataflow_probe.py 3 2 2>&1 | grep -v "^\[elag" > $OUT/dataflow_probe.txt; python tools/r05_dataflow_probe.py 3 6 2>&1 | grep -v "^\[elag" >> $OUT/dataflow_probe.txt
python tools/r05_dataflow_chain.py 2000 >> $OUT/dataflow_probe.txt 2>&1; cat $OUT/dataflow_probe.txt | cut -c1-200
python tools/gs_setup_probe.py > $OUT/gs_setup.txt 2>&1; tail -4 $OUT/gs_setup.txt | cut -c1-300
echo "[6] H(div) x3 kernel stats"; bash tools/profile_bench.sh r05final/hdiv --form 2 --no-cpu-baseline --no-roofline --no-gs --no-other-configs > $OUT/hdiv_kernel_stats.txt 2>&1; tail -2 $OUT/hdiv_kernel_stats.txt | cut -c1-160
echo "[7] the reference's own 2-form list: Hiptmair(L1 Gauss-Seidel, L1 Gauss-Seidel), dataflow on / off"
for m in 1 0; do ELAG_HIP_SWEEP_DATAFLOW=$m timeout -k 10 400 python bench.py --form 2 --smoother "L1 Gauss-Seidel" --no-roofline --no-cpu-baseline --no-other-configs --no-gs > $OUT/hdiv_gs_dataflow_$m.json 2> $OUT/hdiv_gs_dataflow_$m.err; echo "rc $?"; done
echo "[8] the N-rank code path on one rank"; timeout -k 10 400 python bench.py --force-dist --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_multirank_path_1rank.json 2> $OUT/bench_multirank_path_1rank.err; echo "rc $?"
echo "[9] config 4 with its solve"; timeout -k 10 300 python bench.py --form darcy --solve --no-cpu-baseline --no-cpu-solve --steps 20 --warmup 5 > $OUT/bench_darcy_solve.json 2> $OUT/bench_darcy_solve.err; echo "rc $?"
echo "[10] GMRES: a step's Gram-Schmidt chain in one launch -- per link, per step, soak"
(python tools/r05_gmres_chain_probe.py 706000 50; python tools/r05_gmres_chain_probe.py 1400000 40; python tools/r05_gmres_chain_probe.py 30000 50; bash tools/r05_gmres_chain_profile.sh r05final/chain 2>&1 | grep -v "rocprofv3\]\|Opened result") > $OUT/gmres_chain.txt 2>&1; tail -4 $OUT/gmres_chain.txt | cut -c1-220
python tools/r05_gmres_chain_soak.py 1000 > $OUT/gmres_chain_soak.txt 2>&1; tail -3 $OUT/gmres_chain_soak.txt | cut -c1-200
echo "[11] the reference's default example end to end (L1 Gauss-Seidel hierarchy set-up + one PCG solve); the smoother's set-up at scale"
(python tools/r05_gs_time_to_solution.py 2>&1 | grep -v "^\[elag"; python tools/r05_gs_setup_scale.py 4 2>&1 | grep -v "^\[elag" | tail -2; python tools/r05_gs_setup_scale.py 5 2>&1 | grep -v "^\[elag" | tail -2) > $OUT/gs_time_to_solution.txt 2>&1; tail -4 $OUT/gs_time_to_solution.txt | cut -c1-200
echo "[12] one run of the default example, phase by phase (first run of a process and later ones); a PCG pass with and without cg_dir_amul_dot_kernel; a GMRES step of the Darcy solve kernel by kernel"
(python tools/r05_single_run.py "L1 Jacobi"; ELAG_HIP_EXPECTED_CYCLES=16 python tools/r05_single_run.py "L1 Jacobi"; python tools/r05_single_run.py "L1 Gauss-Seidel"; ELAG_HIP_EXPECTED_CYCLES=16 python tools/r05_single_run.py "L1 Gauss-Seidel") 2>&1 | grep -v "^\[elag" > $OUT/single_run.txt; tail -3 $OUT/single_run.txt | cut -c1-220
(python tools/r05_pcg_pass_probe.py 3 256; python tools/r05_pcg_pass_probe.py 3 256; python tools/r05_pcg_pass_probe.py 2 256; bash tools/r05_pcg_pass_profile.sh r05final/pcgprof 3 256 | grep -a "cg_\|spmv_fused\|dot0\|flat_") 2>&1 | grep -v "^\[elag" > $OUT/pcg_pass.txt; cat $OUT/pcg_pass.txt | cut -c1-200
bash tools/r05_darcy_iteration_timeline.sh r05final/dtl > $OUT/darcy_step_timeline.txt 2>&1; tail -26 $OUT/darcy_step_timeline.txt | cut -c1-140
# the raw traces stay on the box: what is merged back is capped at 64 MiB, the summaries above are what profiles/ keeps
find $OUT -type d \( -name kt -o -name tr \) -prune -exec rm -rf {} + ; du -sh $OUT | cut -f1
