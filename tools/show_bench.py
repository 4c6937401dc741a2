import json,sys
for line in sys.stdin:
    if not line.startswith("{"): continue
    d=json.loads(line)
    print(d["config"]["workload"][:90])
    print("   value", d["value"], "device", d.get("device_vcycles_per_s"), "setup_s", d.get("setup_s"), "parity", d.get("parity_rel_err_vs_oracle"))
    print("   solve", json.dumps(d.get("solve"))[:900])
