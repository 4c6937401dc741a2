#!/bin/bash
# A/B of the row-group kernels' lane rule on the GPU box (device ms per cycle of three launch-bound cycles)
for cfg in "4 65536" "4 32768" "2 65536" "8 65536" "4 131072"; do
  set -- $cfg
  echo "== LANE_DIV=$1 MIN_LANES_TOTAL=$2"
  for form in "--form 0" "--form 2" "--form darcy" "--form 0 --smoother Chebyshev"; do
    ELAG_HIP_FLAT_CYCLE=0 ELAG_HIP_LANE_DIV=$1 ELAG_HIP_MIN_LANES_TOTAL=$2 python bench.py $form --no-roofline --no-cpu-baseline --no-gs --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   %-34s device %.4f ms  host-loop %.4f ms  mode %s' % ('$form', d['device_ms_per_step'], d['ms_per_step'], d['config']['launch_mode']))"
  done
done
