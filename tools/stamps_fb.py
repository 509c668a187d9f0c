"""s_memtime stamps inside k_factor_blk (instrumented build, -DHQPKKT_STAMPS) on one dense block
(hqpkkt_debug_factor_block, one workgroup): per panel the phases of thread 0, and for panel 3 the
phases of every wavefront.
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_fb.py [p] [kind]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hqp_amd import _lib  # noqa: E402
from tests import blockcheck as bc  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 160
kind = sys.argv[2] if len(sys.argv) > 2 else "qd"
A = bc.make_block(kind, p, p)
for rep in range(3):
    out = bc.factor_block(A, variant=0, reps=1)
cnt = out["counters"].astype(np.int64)
s = cnt[9:9 + 54]  # FBSTAMP(slot) -> counters[8 + slot] = flags[9 + slot]
npan = (p + 15) // 16
u = lambda a, b: int((b - a) & 0xffffffff)
print(f"{kind} p={p}: {out['ms'] * 1e3:.1f} us per launch (events); shader cycles of thread 0 (s_memtime): "
      f"load {u(s[0], s[1])}, total {u(s[0], s[53])}")
print("panel  elim(or hot)  solve+barrier  test+update  end-barrier   sum")
prev = s[1]
for k in range(min(npan, 10)):
    a = [s[2 + 5 * k], s[3 + 5 * k], s[5 + 5 * k], s[6 + 5 * k]]
    d = [u(prev, a[0]), u(a[0], a[1]), u(a[1], a[2]), u(a[2], a[3])]
    print(f"{k:5d} {d[0]:12d} {d[1]:13d} {d[2]:12d} {d[3]:12d} {sum(d):6d}")
    prev = a[3]
print("tail (M to memory, pivot data):", u(prev, s[53]))
w = (C.c_int * 256)()
_lib.lib().hqpkkt_debug_fb_stamps(w)
w = np.array(w[:], dtype=np.int64).reshape(16, 16)
names = ["tn loaded", "solve phase", "barrier", "test", "elim / update", "publish next", "L11+pivots", "end barrier"]
print("panel 3 per wavefront (ticks between the stamps 0..8):")
print("wave " + " ".join(f"{n[:13]:>13s}" for n in names))
for wv in range(12):
    if w[wv, 0] == 0 and w[wv, 8] == 0:
        continue
    d = [u(w[wv, j], w[wv, j + 1]) for j in range(8)]
    print(f"{wv:4d} " + " ".join(f"{x:13d}" for x in d) + f"   start {u(w[:12, 0].min(), w[wv, 0])}")
