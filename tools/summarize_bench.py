#!/usr/bin/env python3
"""One line per bench.py JSON file: value, step time, kernel launch times, host timers."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:
        print(path, "ERR", e)
        continue
    r = d["roofline"]; k = r["kernels"]; h = d.get("host_seconds", {})
    mol = d.get("molecules_per_chain", {}).get("mean", 0)
    print("%-44s %.3g acc/s  %.3f ms/step acc %.2f | pair %.1f k %.1f commit %.1f us | host gen %.3f sub %.3f wait %.3f res %.3f com %.3f of %.3f s | N %.0f frac %.3f" % (
        path.split("/")[-1][:-5], d["value"], d["ms_per_step"], d["acceptance"], k["pair_sweep"]["avg_launch_us"], k["k_sweep"]["avg_launch_us"],
        k["commit"]["avg_launch_us"], h.get("generate", 0), h.get("submit", 0), h.get("wait", 0), h.get("resolve", 0), h.get("commit", 0),
        d["timed_region_s"], mol, r.get("frac") or 0))
