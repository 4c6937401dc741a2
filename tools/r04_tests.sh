#!/bin/bash
# full GPU test suite + setup probes (first build of a process against repeated builds)
TAG=${1:-r04t}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.txt
tail -3 $OUT/pytest.txt
grep -q "rc 0" $OUT/pytest.txt || exit 1
python tools/setup_probe.py 5 4 > $OUT/setup_k5.txt 2>&1; cat $OUT/setup_k5.txt | tail -4
python tools/setup_probe.py 3 4 > $OUT/setup_k3.txt 2>&1; cat $OUT/setup_k3.txt | tail -4
