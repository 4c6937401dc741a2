#!/usr/bin/env python3
"""The headline fields of bench.py's JSON lines: `python bench.py ... | python tools/show_bench.py` or `python tools/show_bench.py file.json ...`."""
import json, sys
def lines():
    if len(sys.argv) > 1:
        for f in sys.argv[1:]:
            with open(f) as fh:
                yield from fh
    else:
        yield from sys.stdin
for line in lines():
    if not line.startswith("{"): continue
    d = json.loads(line)
    print(d["config"]["workload"][:90])
    print("   value", d["value"], "device", d.get("device_vcycles_per_s"), "setup_s", d.get("setup_s"), "parity", d.get("parity_rel_err_vs_oracle"))
    print("   solve", json.dumps(d.get("solve"))[:900])
