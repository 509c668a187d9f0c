"""tools/fuzz_ip.py's check on a list of case numbers (and ranges a-b): python tools/fuzz_ip_cases.py 193 2536 800-1199 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_ip
cases = []
for a in sys.argv[1:]:
    if "-" in a:
        lo, hi = a.split("-")
        cases += list(range(int(lo), int(hi) + 1))
    else:
        cases.append(int(a))
bad = []
for c in cases:
    st, line = fuzz_ip.check(c)
    if st == "BAD":
        bad.append(c)
        print(line, flush=True)
print(f"{len(cases)} cases: {len(bad)} bad {bad}", flush=True)
