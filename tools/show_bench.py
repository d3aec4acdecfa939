"""Print a bench line (file of JSON lines) in readable pieces (dev tool)."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ex = d.pop("extra", {})
print(json.dumps({k: d.get(k) for k in ("value", "pairs_per_s", "ms_per_step", "roofline", "plan_inclusive", "resident_replay", "end_to_end", "cpu_baseline")}, indent=0)[:4000])
print(json.dumps(ex, indent=0)[:7000])
