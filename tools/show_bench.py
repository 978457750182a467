#!/usr/bin/env python
"""Pretty-print the headline fields of a bench.py JSON line (file argument or stdin)."""
import json, sys
d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
r = d["roofline"]
print("value %.1f %s  ms/step %.3f  one-at-a-time %s" % (d["value"], d["unit"], d["ms_per_step"], d.get("value_one_frame_at_a_time")))
print({k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_calibrated", "frac_timed_region", "frac_timed_region_calibrated", "traffic", "kernel_ms_solo", "kernel_ms_in_flight")})
print("l1_gather", r.get("l1_gather"))
print("hbm", r.get("hbm"))
print("hbm_algorithmic ratio", (r.get("hbm_algorithmic") or {}).get("ratio_to_hbm_peak"), " executed tap bytes", r.get("executed_tap_bytes"))
print("pmc", r.get("pmc"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("single_thread", {}).get("value"))
