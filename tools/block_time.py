"""Latency of the pivot-block kernels on one front (hqpkkt_debug_factor_block): one workgroup, average of 20."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import blockcheck as bc  # noqa: E402

for kind in ("qd", "indef"):
    for p in (48, 80, 100, 128, 150, 160, 192):
        A = bc.make_block(kind, p, p)
        row = [f"{kind:6s} p={p:3d}"]
        for variant, name in ((0, "blk"), (2, "blk16"), (1, "diag")):
            if variant == 1 and p > 128:
                continue
            out = bc.factor_block(A, variant=variant, reps=20)
            err, inv, ok, growth = bc.check_block(A, out)
            c = out["counters"]
            row.append(f"{name}: {out['ms'] * 1e3:7.1f} us (err {err:.1e} inv {inv:.1e} slow {c[3]} 2x2 {c[1]})")
        print("  ".join(row), flush=True)
