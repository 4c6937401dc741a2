#!/bin/bash
# GPU tests, then the default bench line without the PMC passes and the CPU baseline (other_configs: Darcy hierarchy setup with the fused column kernel)
TAG=${1:-r04tb}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.txt
tail -3 $OUT/pytest.txt
grep -q "rc 0" $OUT/pytest.txt || exit 1
timeout -k 10 400 python bench.py --no-traffic --no-cpu-baseline --no-gs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
ELAG_HIP_LU_FUSED_COLUMN=0 timeout -k 10 400 python bench.py --no-traffic --no-cpu-baseline --no-gs --no-roofline > $OUT/bench_unfused.json 2> $OUT/bench_unfused.err; echo "bench rc $?"
