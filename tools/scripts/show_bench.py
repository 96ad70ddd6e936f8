#!/usr/bin/env python3
"""Print the interesting numbers of a bench.py JSON line (argument: the file)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s  ms/step %.2f  hbm frac %.2e  valu frac %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["valu_roofline"].get("frac")))
if "e2e" in d:
    print("e2e", {k: v for k, v in d["e2e"].items() if k != "note"})
for k in ("mixed_alphabet_path", "generic_path", "related_path"):
    if k in d:
        print(k, round(d[k]["value"], 1))
for k, c in d.get("configs", {}).items():
    print("config", k, "kernel_ms", c.get("kernel_ms"), "gcups", c.get("gcups"), "verified", c.get("verified"), "walk", c.get("walk_kernel_ms"))
print("cpu_baseline", d.get("cpu_baseline"))
