"""Latency of k_factor_blk on one well-conditioned block of each size (timing experiments with instrumented builds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import blockcheck as bc  # noqa: E402

for p in [int(a) for a in sys.argv[1:]] or [80, 128, 160, 192]:
    A = bc.make_block("spd", p, p)
    out = bc.factor_block(A, variant=0, reps=20)
    print(f"p={p}: {out['ms'] * 1e3:.1f} us, slow steps {out['counters'][3] // 20}", flush=True)
