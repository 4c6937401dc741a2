#!/bin/bash
# x-cache gates (ELAG_HIP_XC_MIN_REUSE, ELAG_HIP_XC_MIN_NNZ) against config 3 (H(div) x3), config 4 (Darcy x3), headline x3 and the x5 V-cycle
OUT=$PWD/gpurun_out/r04s2m; mkdir -p $OUT
for cfg in "4.0 2097152" "4.0 1048576" "3.0 2097152" "3.0 1048576" "2.0 1048576" "2.0 262144" "3.0 262144"; do
  set -- $cfg
  export ELAG_HIP_XC_MIN_REUSE=$1 ELAG_HIP_XC_MIN_NNZ=$2 ELAG_HIP_XC_VERBOSE=1
  tag=r$1_n$2
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-gs --no-traffic > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python - <<PY
import json
d=json.load(open("$OUT/b_$tag.json"))
o=d["other_configs"]; r=d["roofline"]
print("reuse>=$1 nnz>=$2: headline %.0f | cfg3 %.0f /s | darcy cycle %.0f /s solve %.2f ms | x5 vcycle %.4f ms spmv %.1f us pcg %.2f ms setup %.1f ms" % (d["value"], o["config3_hdiv_hiptmair"]["vcycles_per_s"], o["config4_darcy"]["vcycles_per_s"], o["config4_darcy"]["solve"]["ms_per_solve"], r["vcycle_ms"], r["ms_per_launch"]*1e3, r["pcg"]["ms_per_solve"], r["hierarchy_setup_repeat_s"]*1e3))
PY
done
