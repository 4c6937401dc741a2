#!/bin/bash
TAG=${1:-r04s2c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_sweeps.py tests/test_gpu_blocked.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.txt
tail -30 $OUT/pytest.txt
grep -q "rc 0" $OUT/pytest.txt || exit 1
timeout -k 10 400 python bench.py --no-traffic --no-cpu-baseline --no-gs --no-roofline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
