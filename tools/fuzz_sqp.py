"""The reference's SQP demo (Prg_DID, Hqp_SqpPowell: BASELINE.json configs[0]) over a range of
horizons with every QP solver / KKT plugin pair, reference-only against the same run with the
HIP plugins (oracle/_ref/libhqphost_hip.so).  Usage: python tools/fuzz_sqp.py [kmax ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refapi
ks = [int(a) for a in sys.argv[1:]] or [7, 13, 20, 33, 50, 64, 77, 100, 150, 200, 333, 500]
bad = 0
for k in ks:
    for qp, mat in (("Mehrotra", "RedSpBKP"), ("Mehrotra", "SpBKP"), ("Franke", "RedSpBKP"), ("Franke", "SpBKP")):
        ref = refapi.sqp_did(k, qp, mat, host="hip")
        for qh, mh in ((qp, mat + "Hip"), (qp + "Hip", mat + "Hip")):
            try:
                hip = refapi.sqp_did(k, qh, mh, host="hip")
            except Exception as e:
                bad += 1
                print("EXCEPTION", k, qh, mh, repr(e)[:100], flush=True)
                continue
            ok = hip["rc"] == ref["rc"] and abs(hip["f"] - ref["f"]) <= 1e-6 * max(1.0, abs(ref["f"])) and \
                abs(hip["sqp_iters"] - ref["sqp_iters"]) <= 1
            word = "ok      "
            if not ok and ref["rc"] != 0 and hip["rc"] == ref["rc"]:
                word = "reference not converged either:"  # (sqp_max_iters reached on both sides; the iterates differ)
            elif not ok and hip["rc"] == ref["rc"] == 0 and abs(hip["f"] - ref["f"]) <= 1e-6 * max(1.0, abs(ref["f"])) \
                    and hip["sqp_iters"] < ref["sqp_iters"]:
                # the reference's first QP ends "degenerate"/"suboptimal" next to its solution and the
                # SQP method needs more iterations to recover; ours ends optimal
                word = "fewer SQP iterations than the reference:"
            elif not ok:
                word = "MISMATCH"
                bad += 1
            print(word, k, qh, mh, "rc %d/%d f %.9g/%.9g sqp %d/%d qp %d/%d  %.3f/%.3f s" % (
                hip["rc"], ref["rc"], hip["f"], ref["f"], hip["sqp_iters"], ref["sqp_iters"], hip["qp_iters"], ref["qp_iters"],
                hip["seconds"], ref["seconds"]), flush=True)
print(bad, "bad")
sys.exit(1 if bad else 0)
