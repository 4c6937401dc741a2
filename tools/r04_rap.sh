#!/bin/bash
# RAP-focused GPU call: parity tests that touch the sparse products, kernel stats + timeline (+ SQ counters with $2 = pmc)
TAG=${1:-r04b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.txt
tail -3 $OUT/pytest.txt
grep -q "rc 0" $OUT/pytest.txt || exit 1
bash tools/rap_profile.sh $TAG/rap 5 > $OUT/rap_stats.txt 2>&1
python3 tools/rap_timeline.py $OUT/rap/kt > $OUT/rap_timeline.txt 2>&1
tail -1 $OUT/rap_timeline.txt; head -12 $OUT/rap_stats.txt
if [ "$2" = "pmc" ]; then bash tools/pmc_rap.sh $TAG/pmc 5 > $OUT/rap_pmc.txt 2>&1; fi
python tools/rap_profile.py 5 > $OUT/rap_plain.txt 2>&1; cat $OUT/rap_plain.txt
