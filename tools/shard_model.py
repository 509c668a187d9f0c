"""DESIGN.md section 7's model of ONE system over P ranks, from the pieces measured on one MI355X (the JSON lines of
tools/shard_pieces.py in profiles/r05_slice_products.txt: wall time per stage of what one rank runs - its products,
the control-sized chain, the host's launches and waits of the callback transport - and the bytes of all slots of a stage's
two gathers) and ONE assumption, stated here: a gather runs as direct copies over the xGMI mesh - every rank sends its
slot to each of the other P - 1 over the link it shares with it, all links at once - at `link` GB/s per direction
(76.8 = half of the 153.6 GB/s a link is rated at in both directions; RCCL's ring gathers reach less).  The gather of
the F blocks is static data requested a stage ahead: it counts only where it takes longer than the stage's computing.
No scaling curve has been MEASURED: no multi-GPU node was available in any round.  python tools/shard_model.py [file] [link GB/s]"""
import json, sys
f = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_slice_products.txt"
link = float(sys.argv[2]) if len(sys.argv) > 2 else 76.8
rows = {}
for line in open(f):
    if line.startswith("{"):
        r = json.loads(line)
        rows.setdefault(r["ranks"], []).append(r)
base = rows[1][0]["ms_per_stage"]
print(f"one rank: {base:.3f} ms per stage (K = {rows[1][0]['K']}, nx = {rows[1][0]['nx']}); link {link} GB/s per direction")
print("ranks | computing ms/stage (slowest measured rank) | F blocks MB/link (a stage ahead) | G_xx blocks MB/link | exposed travel ms | total | speed-up | computing alone")
for P in sorted(k for k in rows if k > 1):
    rs = rows[P]
    comp = max(r["ms_per_stage"] for r in rs)
    nx, nu = rs[0]["nx"], rs[0]["nu"]
    xw = 128 * -(-(-(-nx // 128)) // P)
    fg = nx * ((xw + nu + 7) // 8 * 8) * 8.0  # one rank's local block [F_p | F_u]
    per_link = rs[0]["bytes_exchange_factor"] / rs[0]["K"] / P  # one rank's slots of both gathers
    xb = per_link - fg
    t_f, t_x = fg / (link * 1e9) * 1e3, xb / (link * 1e9) * 1e3
    exposed = t_x + max(0.0, t_f - comp)
    total = comp + exposed
    print(f"{P:5d} | {comp:8.3f} | {fg / 1e6:8.1f} | {xb / 1e6:8.1f} | {exposed:6.3f} | {total:6.3f} | x{base / total:4.2f} | x{base / comp:4.2f}")
