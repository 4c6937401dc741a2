#!/bin/bash
for cfg in "0 1" "1 1" "0 0"; do
  set -- $cfg
  export ELAG_HIP_WARM_COPIES=$1 ELAG_HIP_KERNEL_COPIES=$2
  echo "== WARM_COPIES=$1 KERNEL_COPIES=$2"
  python tools/setup_probe.py 5 2 2>&1 | grep build | cut -c1-60
  python tools/setup_probe.py 3 2 2>&1 | grep build | cut -c1-60
  python tools/r04_darcy_setup_probe.py 3 2
done
