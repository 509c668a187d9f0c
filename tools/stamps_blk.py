"""s_memtime stamps inside k_factor_blk (instrumented build, -DHQPKKT_STAMPS) on one dense block:
   HQPKKT_LIB=hqp_amd/libhqpkkt_stamps.so python3 tools/stamps_blk.py [p ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tests import blockcheck as bc  # noqa: E402

for p in [int(a) for a in sys.argv[1:]] or [80, 128, 160]:
    for variant in (0, 2):
        A = bc.make_block("qd", p, p)
        bc.factor_block(A, variant=variant)
        out = bc.factor_block(A, variant=variant)
        s = out["counters"][9:9 + 54].astype(np.int64)
        npan = (p + 15) // 16
        print(f"p {p} variant {variant}: load {s[1] - s[0]}  total {s[53] - s[0]}  ({out['ms'] * 1e3:.1f} us)")
        for k in range(min(npan, 10)):
            a = s[2 + 5 * k: 7 + 5 * k]
            prev = s[1] if k == 0 else s[6 + 5 * (k - 1)]
            print(f"   panel {k}: prologue {a[0] - prev}  solve {a[1] - a[0]}  test + next block row {a[2] - a[1]}  rest of the update | next elimination {a[3] - a[2]}  sync {a[4] - a[3]}")
        import ctypes as C
        from hqp_amd import _lib
        buf = (C.c_int * 256)()
        _lib.lib().hqpkkt_debug_fb_stamps(buf)
        w = np.array(buf[:], dtype=np.int64).reshape(16, 16)[:, :9]
        names = "top | operands of the wavefront | solve / next diagonal block | barrier | test | update / elimination | published | L, pivot data | barrier"
        print("   panel 3 per wavefront, cycles between: " + names)
        for wv in range(16):
            if w[wv].any():
                print(f"      wave {wv:2d}: " + "  ".join(f"{int(w[wv][j + 1] - w[wv][j]):6d}" for j in range(8)))
