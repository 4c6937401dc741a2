#!/bin/bash
# A/B of the flattened cycle's lane rule on the GPU box: device us per cycle for several settings
mkdir -p gpurun_out/r03a
for cfg in "4 4 32768" "8 4 32768" "8 8 32768" "8 2 32768" "2 2 32768" "2 4 32768" "4 3 32768"; do
  set -- $cfg
  echo "== U=$1 LANE_DIV=$2 MIN_LANES=$3"
  ELAG_HIP_FLAT_U=$1 ELAG_HIP_FLAT_LANE_DIV=$2 ELAG_HIP_FLAT_MIN_LANES=$3 ELAG_HIP_FLAT_VERBOSE=1 python tools/headline_probe.py --modes eager --reps 3 2>&1 | grep -E "flattened|device|K=2000"
done
echo "== no flat"
ELAG_HIP_FLAT_CYCLE=0 python tools/headline_probe.py --modes eager --reps 3 2>&1 | grep -E "device|K=2000"
