#!/bin/bash
# config 3 (H(div) x3 Hiptmair) with the x-cache form forced on / heuristic / off
OUT=$PWD/gpurun_out/r04s2l; mkdir -p $OUT
for mode in default 1 0; do
  if [ $mode = default ]; then unset ELAG_HIP_XC; else export ELAG_HIP_XC=$mode; fi
  timeout -k 10 300 python bench.py --form 2 --no-cpu-baseline --no-roofline --no-gs --no-other-configs > $OUT/hdiv_xc_$mode.json 2> $OUT/hdiv_xc_$mode.err
  python - <<PY
import json
d=json.load(open("$OUT/hdiv_xc_$mode.json"))
print("ELAG_HIP_XC=$mode", d["value"], d["ms_per_step"], d.get("config",{}).get("launch_mode"))
PY
done
