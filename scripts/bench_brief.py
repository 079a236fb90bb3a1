"""One line of the figures of a bench.py JSON line read from stdin (A/B runs)."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
it = d["roofline_iteration"]
ex = {k: round(v.get("steps_per_s", v.get("steps_per_s_aggregate", 0)), 1) for k, v in d.items() if isinstance(v, dict) and ("steps_per_s" in v or "steps_per_s_aggregate" in v)}
print(sys.argv[1] if len(sys.argv) > 1 else "", "value", round(d["value"], 1), "iteration us", round(it["us"], 1), "two-sweep fraction", round(it.get("two_sweep_fraction", 0), 2),
      "dominant", d["roofline"]["kernel"][:16], round(d["roofline"]["frac"], 3), ex)
