"""Why the blocked inverse of a free initial state's matrix gave up (introspection 32 of the library): the cases of
FUZZ_X0=1 tools/fuzz_bigstage.py, STAGED engine only.  python tools/x0_why.py [cases] [seed0]"""
import os
import struct
import sys

os.environ["FUZZ_X0"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_bigstage as fb
from hqp_amd import ipmatrix


def f32(i):
    return struct.unpack("f", struct.pack("i", int(i)))[0]


n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fell = 0
for case in range(s0, s0 + n):
    prog, st, tag = fb.make_case(case)
    S = ipmatrix.IpLQDOCP()
    try:
        S.init(prog)
        S.factor(prog, st[0], st[1])
    except ipmatrix.KktError as e:
        print(tag, "raised", e)
        continue
    w = list(S.debug(32))
    if w[1]:
        fell += 1
        print(tag, "| gave up at block", w[2], "|K_jj| %.2e |K_jj^-1| %.2e" % (f32(w[3]), f32(w[4])), "check %.2e" % f32(w[5]), flush=True)
print(f"{fell} of {n} fell back")
