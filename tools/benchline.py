"""Condense bench.py's JSON line (stdin) to ms_per_step + per-class kernel times."""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("ms_per_step", d["ms_per_step"], {k: round(v, 4) for k, v in d.get("kernel_ms_per_step", {}).items()})
