#!/bin/bash
# ms per launch of the x-cache SpMV on the cube456 x5 fine-level operator for a range of window sizes (entries per workgroup), each in
# its own bench.py --roofline-only run with the form forced (ELAG_HIP_XC=1 ELAG_HIP_XC_WINDOW=<w>); windows above 2048 run 512 threads.
# usage: tools/xc_window_scan.sh [windows...]
W=${@:-1024 1280 1536 1664 1920 2048 2176 2304 2432 2560 2816 3072 3584 4096}
for w in $W; do
  ELAG_HIP_XC=1 ELAG_HIP_XC_WINDOW=$w python3 bench.py --roofline-only --roofline-form xc 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']
print('window %5d  %7.2f us  %7.1f GB/s  %.4f of 8 TB/s   parity %.1e' % (d['window'], d['ms_per_launch']*1e3, d['achieved_GBs'], d['achieved_GBs']/8000, d['parity_roofline_rel_err']))"
done
