#!/bin/bash
# round 4, first GPU call: tests, RAP kernel stats + timeline + SQ counters, the kernel-argument default of a C++ host
OUT=$PWD/gpurun_out/r04a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.txt
tail -3 $OUT/pytest.txt
bash tools/rap_profile.sh r04a/rap 5 > $OUT/rap_stats.txt 2>&1
python3 tools/rap_timeline.py $OUT/rap/kt > $OUT/rap_timeline.txt 2>&1
tail -1 $OUT/rap_timeline.txt; head -16 $OUT/rap_stats.txt
bash tools/pmc_rap.sh r04a/pmc 5 > $OUT/rap_pmc.txt 2>&1
hipcc -O2 --offload-arch=gfx950 tools/kernarg_probe.cpp -Iinclude -Lparelag_amd -lelag_hip -Wl,-rpath,$PWD/parelag_amd -o /tmp/kernarg_probe 2> $OUT/kernarg_build.txt
( env -u HIP_FORCE_DEV_KERNARG /tmp/kernarg_probe; HIP_FORCE_DEV_KERNARG=0 /tmp/kernarg_probe; HIP_FORCE_DEV_KERNARG=1 /tmp/kernarg_probe ) > $OUT/kernarg.txt 2>&1
cat $OUT/kernarg.txt
python tools/setup_probe.py 5 4 > $OUT/setup_k5.txt 2>&1; tail -2 $OUT/setup_k5.txt
