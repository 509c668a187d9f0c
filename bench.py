#!/usr/bin/env python3
"""Benchmark of the MI355X KKT path: KKT factor+solve per second (fp64).

    python bench.py --gpus N --steps K --warmup W [--workload c4|c2]

A "step" is one pass of the hot path over one KKT system whose data are already
resident in HBM: Hqp_IpMatrix::factor followed by Hqp_IpMatrix::solve (one solve +
iterative refinement to mat_eps), i.e. what one interior-point iteration of the reference
asks of its plugin besides extra right-hand sides (hqp/Hqp_IpsMehrotra.C:527-530).

Workload at N=1 (default, --workload c4): the configuration BASELINE.json's metric is quoted
on, the 10^6-variable DOCP = configs[3] / SURVEY.md 8(d) "C4": synthetic multistage LQ
optimal control QP, K = 200 stages of nx = 5000 states and nu = 50 controls (n = 1 015 000,
me = 1 005 000, m = 20 000), dense random stable fx, dense fu, x_0 fixed, box bounds on the
controls; plugin LQDOCP = the STAGED engine (hqpkkt_analyze_staged: dynamics handed over as
dense blocks).  --workload c2: configs[1], the synthetic banded KKT system of dim 10^5 through
the full-system engine (round 1's headline; its rate is also carried as an extra of the c4
line).  N>1: one process per GPU, every rank factors+solves its own system (different seed):
independent KKT systems shard with no data-path collective ("weak"); --one-system: ONE system
over the ranks ("strong").

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 MFMA peak
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md


class Watchdog:
    """N > 1: every phase that can wait for another rank (rendezvous, communicator creation, the first collectives)
    runs under a deadline.  A rank that is still inside when it expires says where and leaves with status 86:
    torchrun then ends the other ranks, so that a hang becomes a failed run with a message within minutes (the
    parent in launch_ranks may then try once more with the other transport).  Never re-execs anything."""

    def __init__(self, rank):
        self.rank, self._t = rank, None

    def arm(self, seconds, what):
        import threading
        self.disarm()

        def fire():
            print(f"bench: rank {self.rank} still in '{what}' after {seconds} s - giving up (exit 86)", file=sys.stderr, flush=True)
            os._exit(86)

        self._t = threading.Timer(seconds, fire)
        self._t.daemon = True
        self._t.start()

    def disarm(self):
        if self._t is not None:
            self._t.cancel()
            self._t = None


def kernel_source_sha16():
    """Fingerprint of the STAGED engine's kernel sources: a counter-derived figure (roofline.traffic) is quoted only
    when the profile it comes from was collected on the same sources."""
    import hashlib
    h = hashlib.sha256()
    for name in ("staged.hip.h", "staged_host.hip.h"):
        with open(os.path.join(ROOT, "hqp_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def class_work(struct, rank=None):
    """Algorithmic work per numeric factorisation, split by kernel class, from the
    symbolic structure (p pivots, b border rows per supernode); flops count a
    multiply-add as 2, the Schur update only its lower triangle.  With ``rank`` set
    (one system sharded over the ranks) only the supernodes that rank factors."""
    p = struct["npiv"].astype(np.float64)
    b = struct["nborder"].astype(np.float64)
    if rank is not None:
        mine = (struct["node_owner"] == rank) | (struct["node_owner"] < 0)
        p, b = p[mine], b[mine]
    return {
        # LDL' of the pivot block (p^3/3) + explicit inverse of the unit lower L11 (p^3/3)
        "factor_diag": {"flops": float((2 * p ** 3 / 3.0).sum()), "bytes": float((8 * 2 * p * p).sum())},
        # X = A21 P' M' (b p^2) ; reads A21 and M, writes X and L21
        "panel_solve": {"flops": float((b * p * p).sum()), "bytes": float((8 * (3 * b * p + p * p / 2)).sum())},
        # U = (children's blocks, pulled) - L21 X': reads L21, X and the children's entries, writes U once
        "schur_update": {"flops": float((b * b * p).sum()), "bytes": float((8 * (2 * b * p + b * b)).sum())},
        # one sweep over the factor: M (lower) and L21 are streamed once
        "solve_fwd": {"flops": float((p * p + 2 * b * p).sum()), "bytes": float((8 * (p * p / 2 + b * p)).sum())},
        "solve_bwd": {"flops": float((p * p + 2 * b * p).sum()), "bytes": float((8 * (p * p / 2 + b * p)).sum())},
    }


def cpu_baseline(prog, state, budget_s=25.0):
    """Reference CPU path timed on this box's host cores (one core: the path is
    single-threaded, hqp/spBKP.C).  kind "reference" = the reference's own
    Hqp_IpSpBKP built from its sources (oracle/_ref); "port" = our C oracle on a
    reduced sample when the reference build cannot be loaded."""
    try:
        from oracle import refapi
        have_ref = refapi.available()
    except Exception:
        have_ref = False
    if have_ref:
        R = refapi.RefIpMatrix("SpBKP")
        R.init(prog)
        t_used, ts = R.t_init, []
        while True:
            R.factor(state[0], state[1])
            _d, res = R.solve(*state)
            ts.append(R.t_factor + R.t_solve)
            t_used += ts[-1]
            if len(ts) >= 3 or t_used + ts[-1] > budget_s:
                break
        t = float(np.median(ts))
        return {"value": 1.0 / t, "unit": "KKT factor+solve/s", "cores": 1, "kind": "reference",
                "sample": f"{len(ts)} x (Hqp_IpSpBKP::factor + Hqp_IpMatrix::solve) on the same C2 system "
                          f"after one init ({R.t_init:.2f} s, not counted); median {t:.3f} s; residual {res:.2e}",
                "init_s": R.t_init, "factor_plus_solve_s": t}
    from hqp_amd import problems
    from oracle import oracleapi
    small = problems.banded_qp(1600, 32, 12345)
    st = problems.ip_state(small, 1)
    O = oracleapi.OracleIpMatrix("SpBKP")
    O.init(small)
    t0 = time.perf_counter()
    O.factor(st[0], st[1])
    O.solve(*st)
    t = time.perf_counter() - t0
    return {"value": 1.0 / t, "unit": "KKT factor+solve/s", "cores": 1, "kind": "port",
            "sample": "1 x factor+solve of a REDUCED-SIZE system (n=1600, b=32, KKT dim 4000) with the dense-storage "
                      "C oracle; oracle/_ref not loadable on this box"}


def ip_iterations(K=2000):
    """Second half of BASELINE.json's metric, IP-iterations/s: the REFERENCE's own
    Hqp_IpsMehrotra (oracle/_ref/libhqphost_hip.so, built from the reference sources)
    solving the double-integrator-like QP of config C3 once with the reference plugin
    and once with ours (shim/Hqp_IpSpBKPHip.C -> C ABI -> HIP, host pointers as in a
    real HQP run).  None when that library did not travel to this box."""
    try:
        from oracle import refapi
        if not refapi.host_available("hip"):
            return None
        from hqp_amd import problems
        prog = problems.did_like_qp(K)
        out = {"workload": f"C3-like Prg_DID structure K={K}: n={prog.n} me={prog.me} m={prog.m}, Hqp_IpsMehrotra, "
                           "plugin RedSpBKP (reference default) vs RedSpBKPHip"}
        refapi.ip_solve(prog, "Mehrotra", "RedSpBKPHip", host="hip")  # warm-up (analysis upload, graphs)
        for key, mat in (("reference_cpu", "RedSpBKP"), ("hip", "RedSpBKPHip")):
            r = refapi.ip_solve(prog, "Mehrotra", mat, host="hip")
            out[key] = {"iters": r["iters"], "result": r["result"], "seconds": r["seconds"],
                        "ip_iters_per_s": r["iters"] / r["seconds"] if r["seconds"] > 0 else None}
        # the same loop device-resident (hqpkkt_mehrotra: our restatement of the reference's
        # Mehrotra solver with all vector work on the GPU; cold start, second of two runs)
        from hqp_amd import ipmatrix
        M = ipmatrix.IpRedSpBKP()
        M.init(prog)
        M.mehrotra(prog)
        _x, _y, _z, _w, info = M.mehrotra(prog)
        out["hip_device_resident"] = {"iters": info["iters"], "result": info["result"],
                                      "seconds": info["ms_total"] * 1e-3, "factorisations": info["n_factor"],
                                      "solves": info["n_solve"],
                                      "ip_iters_per_s": info["iters"] / (info["ms_total"] * 1e-3)}
        # optional (hqpkkt_opts.amalgamation, off by default): separators absorb their child
        # separators, a third of the tree levels above the leaves
        M = ipmatrix.IpRedSpBKP(amalgamation=True)
        M.init(prog)
        M.mehrotra(prog)
        _x, _y, _z, _w, info = M.mehrotra(prog)
        out["hip_device_resident_amalgamated"] = {"iters": info["iters"], "result": info["result"],
                                                  "seconds": info["ms_total"] * 1e-3,
                                                  "ip_iters_per_s": info["iters"] / (info["ms_total"] * 1e-3)}
        if K <= 4000:
            # the reference's other IP solver (Hqp_IpsFranke, the default of Hqp_SqpSolver): one
            # factor + one solve per iteration; reference alone vs hqpkkt_franke
            r = refapi.ip_solve(prog, "Franke", "RedSpBKP", host="hip", max_iters=400)
            out["franke_reference_cpu"] = {"iters": r["iters"], "result": r["result"], "seconds": r["seconds"],
                                           "ip_iters_per_s": r["iters"] / r["seconds"] if r["seconds"] > 0 else None}
            M = ipmatrix.IpRedSpBKP()
            M.init(prog)
            M.franke(prog, max_iters=400)
            _x, _y, _z, _w, info = M.franke(prog, max_iters=400)
            out["franke_device_resident"] = {"iters": info["iters"], "result": info["result"],
                                             "seconds": info["ms_total"] * 1e-3,
                                             "ip_iters_per_s": info["iters"] / (info["ms_total"] * 1e-3)}
        return out
    except Exception as e:  # never let the secondary measurement break the bench line
        return {"error": str(e)}


def concurrent_systems(prog, state, cls, local_rank, steps, counts=(2, 4)):
    """Extra information, never `value`: aggregate factor+solve/s when SEVERAL independent
    KKT systems of the bench's size are in flight on the one GPU (one handle + stream + host
    thread each; scenario trees, the QPs of separate SQP runs).  A single C2 system leaves
    most of the chip idle (its upper tree levels are a handful of workgroups), so the
    aggregate rate says how much of that idle time other systems can use."""
    import threading
    import torch
    out = {}
    try:
        mats, vecs = [], []
        for k in range(max(counts)):
            m = cls(device=local_rank, device_vectors=True)
            m.init(prog)
            dev = [torch.as_tensor(a).cuda() for a in state]
            d = [torch.zeros(q, dtype=torch.float64, device="cuda") for q in (prog.n, prog.me, prog.m, prog.m)]
            m.factor(prog, dev[0], dev[1])
            m.solve(prog, *dev, *d)
            mats.append(m), vecs.append((dev, d))
        for c in counts:
            def work(i):
                dev, d = vecs[i]
                for _ in range(steps):
                    mats[i].factor(prog, dev[0], dev[1])
                    mats[i].solve(prog, *dev, *d)
            th = [threading.Thread(target=work, args=(i,)) for i in range(c)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            out[str(c)] = c * steps / (time.perf_counter() - t0)
    except Exception as e:  # never let the secondary measurement break the bench line
        out["error"] = str(e)
    return out


# ------------------------------------------------------------------ C4: the 10^6-variable DOCP
def c4_dense(K, nx, nu, seed=0, device="cuda"):
    """SURVEY.md 8(d) "C4" generated on the device: fx dense random with spectral radius ~0.9
    (a different block for every stage), fu dense random, Q = diag(1 on states, 0.1 on
    controls), x_0 fixed, -1 <= u <= 1.  Returns a problems.DenseDocp (F blocks = torch tensors)."""
    import torch
    from hqp_amd import problems
    g = torch.Generator(device=device).manual_seed(seed)
    nz = nx + nu
    F = []
    for _k in range(K):
        blk = torch.empty((nx, nz), dtype=torch.float64, device=device)
        blk.uniform_(-1.0, 1.0, generator=g)
        blk[:, :nx] *= 0.9 / np.sqrt(nx / 3.0)
        F.append(blk)
    n = K * nz + nx
    qd = np.ones(n)
    qd.reshape(-1)[:K * nz].reshape(K, nz)[:, nx:] = 0.1
    Q = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), qd)
    E = (np.arange(nx + 1, dtype=np.int32), np.arange(nx, dtype=np.int32), np.ones(nx))
    ucols = (np.arange(K)[:, None] * nz + nx + np.arange(nu)[None, :]).ravel()
    cols = np.concatenate([ucols, ucols]).astype(np.int32)
    vals = np.concatenate([np.ones(ucols.size), -np.ones(ucols.size)])
    C = (np.arange(cols.size + 1, dtype=np.int32), cols, vals)
    return problems.DenseDocp([nx] * (K + 1), [nu] * K, Q, E, C, F, nx, cols.size)


def c4_qp_vectors(dq, nx, nu, seed=7):
    """c, b, d of the interior-point run on the C4 structure.  The state cost reaches every control through the dense
    f_u (effective control Hessian ~ nx / 3), so with unit bounds no bound is ever active and the QP is solved in three
    iterations; the bounds |u| <= 1.2 / (nx / 3) are of the size of the unconstrained controls instead: a third to a
    half of the controls end at a bound (the bench line and the test report the fraction they found)."""
    import torch
    n, me, m = dq.dims
    gq = torch.Generator(device="cuda").manual_seed(seed)
    dq.c = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-0.5, 0.5, generator=gq)
    dq.b = torch.zeros(me, dtype=torch.float64, device="cuda")
    dq.b[me - nx:] = torch.empty(nx, dtype=torch.float64, device="cuda").uniform_(-1.0, 1.0, generator=gq)
    dq.d = torch.full((m,), 1.2 / (nx / 3.0), dtype=torch.float64, device="cuda")
    return dq


def c4_kkt_norms(F, K, nx, nu, c, b, d, x, y, z, w):
    """Max-norms of the KKT conditions of the C4 QP (Q = diag(1 on states, 0.1 on controls), dynamics blocks F, x_0 fixed,
    |u| <= d) at (x, y, z, w): stationarity Q x + c - A'y - C'z, equalities A x + b, inequalities C x + d - w,
    complementarity z .* w, smallest z / w.  Conventions of hqp/Hqp_IpsMehrotra.C:27-31."""
    import torch
    nz = nx + nu
    X = x[:K * nz].view(K, nz)
    xs = torch.cat([X[:, :nx], x[K * nz:].view(1, nx)])      # x_0 .. x_K
    ydyn, yE = y[:K * nx].view(K, nx), y[K * nx:]
    qd = torch.ones_like(x)
    qd[:K * nz].view(K, nz)[:, nx:] = 0.1
    stat = qd * x + c
    ax = torch.empty_like(y)
    for k in range(K):
        ax[k * nx:(k + 1) * nx] = F[k] @ X[k] - xs[k + 1]
        stat[k * nz:(k + 1) * nz] -= F[k].T @ ydyn[k]
        stat[(k + 1) * nz:(k + 1) * nz + nx] += ydyn[k]      # the -1 of row block k in column block x_{k+1}
    ax[K * nx:] = xs[0]
    stat[:nx] -= yE
    ku = K * nu
    U = X[:, nx:].reshape(-1)
    Su = stat[:K * nz].view(K, nz)[:, nx:]
    Su -= (z[:ku] - z[ku:]).view(K, nu)
    cxd = torch.cat([U, -U]) + d
    return {"stationarity": float(stat.abs().max()), "equalities": float((ax + b).abs().max()),
            "inequalities": float((cxd - w).abs().max()), "complementarity": float((z * w).abs().max()),
            "min_z": float(z.min()), "min_w": float(w.min()), "active_fraction": float((w < z).double().mean())}


def c4_program(K, nx, nu, seed=0):
    """The same QP family in CSR form (Hqp_Docp's layout) for the CPU reference: what
    Hqp_IpLQDOCP::init / factor / step are timed on (hqp_amd.problems.c4_docp_csr)."""
    from hqp_amd import problems
    return problems.c4_docp_csr(K, nx, nu, seed)


def _ref_time(args):
    """One process: median factor+solve time of a reference plugin (Hqp_IpLQDOCP / Hqp_IpSpBKP) on c4_program(K, nx, nu)."""
    K, nx, nu, reps = args[:4]
    kind = args[4] if len(args) > 4 else "LQDOCP"
    from hqp_amd import problems
    from oracle import refapi
    prog = c4_program(K, nx, nu, seed=1)
    st = problems.ip_state(prog, 1)
    R = refapi.RefIpMatrix(kind)
    R.init(prog)
    ts, res = [], None
    for _ in range(reps):
        R.factor(st[0], st[1])
        _d, res = R.solve(*st)
        ts.append(R.t_factor + R.t_solve)
    return float(np.median(ts)), float(res)


_ref_lqdocp_time = _ref_time


def cpu_baseline_c4(K, nx, nu):
    """The reference's own plugins (oracle/_ref, built from the reference's sources) timed on this box's host cores on a
    BOUNDED sample of the workload, as SURVEY.md 8(d) C4 prescribes.  The recursion's cost is linear in the number of
    stages (measured: K = 2 -> 4 at nx = 1000 takes 1.94 x), so the sample is a SLICE of the workload at the widest stages
    the budget allows - K = 2 stages of the same QP family at nx = 1000 and nx = 2000 (about 3 s and 20 s on one core;
    the full nx = 5000 needs ~10^14 flops of Meschach's triple-loop m_mlt: hours) - and `value` is
    1 / (K x the per-stage time at nx = 2000 x 2.5^e) with the exponent e of those two samples: an extrapolation over
    2.5 x in nx (until round 5 the samples were K = 200 at nx <= 400 - 12.5 x - and their exponent 2.28 flattered the CPU:
    a slice at the FULL width, K = 2 at nx = 5000, measured once in the build container, is in profiles/r05_ref_full_width.jsonl).
    The small-size table (K stages at nx = 50, 100, 200) is kept for the size sweep of `staged_small_sizes`; Hqp_IpSpBKP,
    the plugin north_star names as the comparator (its band grows with nx: 13.7 s per factorisation at nx = 200 already),
    at nx = 50, 100.  One core (the path is single-threaded) plus the aggregate of 8 instances on 8 cores."""
    try:
        from oracle import refapi
        have_ref = refapi.available()
    except Exception:
        have_ref = False
    cores = os.cpu_count()
    if not have_ref:
        from hqp_amd import problems
        from oracle import oracleapi
        small = c4_program(12, 12, 3, seed=1)
        st = problems.ip_state(small, 1)
        O = oracleapi.OracleIpMatrix("SpBKP")
        O.init(small)
        t0 = time.perf_counter()
        O.factor(st[0], st[1])
        O.solve(*st)
        t = time.perf_counter() - t0
        return {"value": 1.0 / t, "unit": "KKT factor+solve/s", "cores": 1, "kind": "port", "host_cores": cores,
                "sample": "1 x factor+solve of a REDUCED-SIZE multistage QP (K=12, nx=12, nu=3) with the dense-storage C oracle of "
                          "the full system; oracle/_ref not loadable on this box; NOT extrapolated"}

    def fit(sizes, times):
        lx, lt = np.log(np.asarray(sizes, float)), np.log(np.asarray(times, float))
        expo = float(np.polyfit(lx, lt, 1)[0])
        local = float(np.log(times[-1] / times[-2]) / np.log(sizes[-1] / sizes[-2])) if len(sizes) > 1 else expo
        return {"sizes": list(sizes), "seconds": [float(t) for t in times], "exponent_fit": expo, "exponent_last_two": local,
                "extrapolated_fit_s": float(times[-1] * (nx / sizes[-1]) ** expo),
                "extrapolated_cubic_s": float(times[-1] * (nx / sizes[-1]) ** 3)}

    sizes_l, reps_l = (50, 100, 200), (5, 5, 3)
    tl = [_ref_time((K, s_, nu, r_))[0] for s_, r_ in zip(sizes_l, reps_l)]
    lq = fit(sizes_l, tl)
    ks, wide = 2, (1000, 2000)
    if nx >= wide[0]:
        # the slice at wide stages: K = 2 (per-stage time = half), nx = 1000 and 2000; the samples' power law in BOTH directions
        # (between the samples that is an interpolation, beyond nx = 2000 an extrapolation)
        tw = [_ref_time((ks, s_, nu, 1))[0] / ks for s_ in wide]
        expo = float(np.log(tw[1] / tw[0]) / np.log(wide[1] / wide[0]))
        per_stage = float(tw[1] * (nx / wide[1]) ** expo)
        per_stage_cubic = float(tw[1] * (nx / wide[1]) ** 3)
        how = (f"EXTRAPOLATED over {nx / wide[1]:.1f}x in nx" if nx > wide[1] else "INTERPOLATED between the samples") + \
              f" to nx={nx} with the samples' exponent {expo:.2f}"
        # Second reading: the one slice at the FULL width that was ever measured (build container, one core of another
        # machine: profiles/r05_ref_full_width.jsonl - K = 2 at nx = 2000: 9.305 s per stage, at nx = 5000: 324.04 s per
        # stage, i.e. an exponent of 3.87 over that factor 2.5, where Meschach's m_mlt leaves the caches) carried over
        # to THIS host by the ratio of the two, applied to this host's own nx = 2000 sample.
        fw_expo = float(np.log(324.04 / 9.305) / np.log(2.5))
        per_stage_fw = float(tw[1] * (nx / wide[1]) ** fw_expo) if nx > wide[1] else per_stage
        lq.update({"slice_stages": ks, "slice_nx": list(wide), "slice_seconds_per_stage": tw, "slice_exponent": expo,
                   "full_width_exponent_build_container": fw_expo})
        sample = (f"Hqp_IpLQDOCP::factor + Hqp_IpMatrix::solve on a SLICE of the workload: K={ks} stages of the same QP family (nu={nu}) at "
                  + ", ".join(f"nx={s_}: {t_:.2f} s per stage" for s_, t_ in zip(wide, tw))
                  + f"; the cost is linear in the stages; {how}: {per_stage:.0f} s per stage x K={K} = {K * per_stage:.0f} s per "
                    f"factor+solve (used for `value`: the reading that favours the CPU); with nx^3: {K * per_stage_cubic:.0f} s; with the exponent "
                    f"{fw_expo:.2f} of the one full-width slice measured in the build container (profiles/r05_ref_full_width.jsonl): "
                    f"{K * per_stage_fw:.0f} s")
    else:
        # a small workload (--nx below the wide samples): the K-stage samples at nx = 50, 100, 200 and their power law in both
        # directions; the wide samples (about 25 s of CPU) are not run
        tw, expo = [], lq["exponent_fit"]
        per_stage = float(tl[-1] / K * (nx / sizes_l[-1]) ** expo)
        per_stage_cubic = float(tl[-1] / K * (nx / sizes_l[-1]) ** 3)
        per_stage_fw = per_stage
        inside = sizes_l[0] <= nx <= sizes_l[-1]
        sample = (f"Hqp_IpLQDOCP::factor + Hqp_IpMatrix::solve on the workload's own K={K} stages (nu={nu}) at "
                  + ", ".join(f"nx={s_}: {t_:.2f} s" for s_, t_ in zip(sizes_l, tl))
                  + f"; {'INTERPOLATED' if inside else 'EXTRAPOLATED'} to nx={nx} with their fitted exponent {expo:.2f}: "
                    f"{K * per_stage:.2f} s per factor+solve")
    full_s, full_cubic_s, full_fw_s = K * per_stage, K * per_stage_cubic, K * per_stage_fw
    lq.update({"extrapolated_fit_s": full_s, "extrapolated_cubic_s": full_cubic_s, "extrapolated_full_width_exponent_s": full_fw_s})
    out = {"value": 1.0 / full_s, "unit": "KKT factor+solve/s", "cores": 1, "kind": "reference", "host_cores": cores,
           "sample": sample, "value_full_width_exponent": 1.0 / full_fw_s,
           "lqdocp": lq, "measured": {"nx100_s": tl[1], "nx200_s": tl[2], "exponent": lq["exponent_fit"]},
           "extrapolated_s": full_s, "extrapolated_full_width_exponent_s": full_fw_s}
    try:  # Hqp_IpSpBKP, the comparator north_star names (full KKT system, RCM band of ~3 nx)
        sizes_s = (50, 100)
        ts_ = [_ref_time((K, s_, nu, 2, "SpBKP"))[0] for s_ in sizes_s]
        out["spbkp"] = fit(sizes_s, ts_)
        out["spbkp"]["note"] = ("Hqp_IpSpBKP::factor + solve on the same QPs; its work grows ~nx^3 (band ~3 nx), memory ~N*band: the nx=5000 system "
                                "would need ~340 GB of factor (SURVEY.md 6) - extrapolation only")
    except Exception as e:
        out["spbkp"] = {"error": str(e)}
    try:  # 8 independent instances on the host's cores: aggregate rate at the nx=100 sample
        import subprocess
        inst = min(8, cores or 1)
        code = ("import sys; sys.path.insert(0, %r); import bench; t, r = bench._ref_time((%d, 100, %d, 3)); print(t)"
                % (ROOT, K, nu))
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(inst)]
        ts = [float(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in procs]
        wall = time.perf_counter() - t0
        tm = float(np.median(ts))
        out["aggregate_instances"] = {"instances": inst, "median_s_per_instance_nx100": tm, "slowdown_vs_alone": tm / tl[1],
                                      "aggregate_value_extrapolated": inst / (lq["extrapolated_fit_s"] * tm / tl[1]), "wall_s": wall}
    except Exception as e:
        out["aggregate_instances"] = {"error": str(e)}
    return out


def mesh_kkt_sizes(local_rank):
    """Extra information: BASELINE configs[4]'s stand-in (mesh-structured sparse QP, plugin RedSpBKP, the tree of the
    graph's own dissection: hqpkkt_opts.ordering 2): KKT factor+solve at 9e4 and 1e6 variables."""
    import torch
    from hqp_amd import ipmatrix, problems
    out = {}
    try:
        for g in (300, 1000, -1, -2):
            # (-1: no mesh - 10^5 variables, 21 entries per row of Q, 1000 random far couplings: problems.banded_long_range_qp;
            # -2: the 10^6-cell mesh with 1 % = 10 000 couplings between distant cells: configs[4]'s "random sparse" at full size)
            prog = (problems.grid_sparse_qp(g, g) if g > 0 else problems.banded_long_range_qp(100000, 10, 1000) if g == -1
                    else problems.grid_sparse_qp(1000, 1000, seed=5, long_range=10000))
            st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 1, 1.0)]
            M = ipmatrix.IpRedSpBKP(device=local_rank, device_vectors=True, ordering=2)
            t0 = time.perf_counter()
            M.init(prog)
            init_s = time.perf_counter() - t0
            d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                M.factor(prog, st[0], st[1])
                res = M.solve(prog, *st, *d)
                ts.append(time.perf_counter() - t0)
            s = M.stats()
            out[f"{g}x{g}" if g > 0 else "band21_far1000_n1e5" if g == -1 else "1000x1000_far10000_n1e6"] = {"variables": prog.n, "kkt_dim": s["dim"], "ms_per_factor_solve": 1e3 * float(np.median(ts[1:])),
                               "residual": res, "init_s": init_s, "flops_factor": s["flops_factor"], "tree_levels": s["n_levels"],
                               "max_front": s["max_front"]}
            del M
    except Exception as e:
        out["error"] = str(e)
    return out


def staged_small_sizes(local_rank):
    """Extra information: the STAGED engine at the sizes the reference's Hqp_IpLQDOCP is timed at in
    SURVEY.md section 6 (K=200, nu=10, CSR hand-over), next to the full-system engine on the same QPs."""
    import torch
    from hqp_amd import ipmatrix, problems
    out = {}
    try:
        for nx in (50, 100, 200, 400):
            prog = problems.lq_docp(200, nx, 10, seed=11)
            st = [torch.as_tensor(a).cuda() for a in problems.ip_state(prog, 5, 1.0)]
            row = {}
            for name, cls in (("staged", ipmatrix.IpLQDOCP), ("full_engine", ipmatrix.IpLQDOCPFull)):
                M = cls(device=local_rank, device_vectors=True)
                M.init(prog)
                d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]
                ts = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    M.factor(prog, st[0], st[1])
                    res = M.solve(prog, *st, *d)
                    ts.append(time.perf_counter() - t0)
                row[name] = {"ms": 1e3 * float(np.median(ts[1:])), "residual": res}
                del M
            out[f"nx{nx}"] = row
    except Exception as e:
        out["error"] = str(e)
    return out


def bench_c4(args):
    import torch
    from hqp_amd import dist as kdist
    rank, local_rank, world = kdist.env_world()
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dog = Watchdog(rank)
    if world > 1:
        dog.arm(args.watchdog, "process group rendezvous")
    if args.share_gpu and world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group(args.backend)
    else:
        kdist.init(args.backend)
    dog.disarm()
    from hqp_amd import ipmatrix
    K, nx, nu = args.stages, args.nx, args.nu
    # N > 1: ONE system over the ranks (strong scaling) unless --replicas
    one = world > 1 and not args.replicas
    dq = c4_dense(K, nx, nu, seed=0 if one else rank)
    n, me, m = dq.dims
    shard, transport, comm_ranks = None, None, None
    if one:
        # every rank must hold the SAME system: the device generators should agree (same seed, same kind of
        # GPU); if a checksum says otherwise, rank 0's blocks are broadcast
        import torch.distributed as tdist
        dog.arm(args.watchdog, "first collectives / communicator of the sharded system")
        chk = torch.stack([blk.sum() for blk in dq.F]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        if args.backend != "nccl":
            lo, hi = lo.cpu(), hi.cpu()
        tdist.all_reduce(lo, op=tdist.ReduceOp.MIN)
        tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)
        if float(lo) != float(hi):
            for blk in dq.F:
                if args.backend == "nccl":
                    tdist.broadcast(blk, src=0)
                else:
                    t = blk.cpu()
                    tdist.broadcast(t, src=0)
                    blk.copy_(t)
        if args.backend == "nccl" and args.transport == "rccl":
            try:  # libhqpkkt_rccl.so: ncclAllGather in the handle's stream
                shard, transport = kdist.RcclShard(rank, world, local_rank), "libhqpkkt_rccl (RCCL, stream-ordered)"
                comm_ranks = shard.comm_ranks  # ncclCommCount of the communicator the collectives run on
            except Exception as e:  # fall back to torch.distributed's collectives behind the callback
                print(f"bench: RcclShard failed ({e}); using the torch.distributed callback", file=sys.stderr)
                shard = None
            # the choice of transport is one decision of ALL ranks: a rank that fell back alone would wait in a
            # collective the others never enter
            okf = torch.tensor([1.0 if shard is not None else 0.0], device="cuda")
            tdist.all_reduce(okf, op=tdist.ReduceOp.MIN)
            if float(okf) == 0.0 and shard is not None:
                shard.close()
                shard = None
        if shard is None:
            shard = (rank, world, kdist.make_exchange(rank, local_rank))
            transport = f"torch.distributed ({args.backend}) behind the exchange callback"
            import torch.distributed as tdist
            comm_ranks = tdist.get_world_size()
        dog.disarm()
    mat = ipmatrix.IpLQDOCP(device=local_rank, device_vectors=True, shard=shard)
    t0 = time.perf_counter()
    mat.init_dense(dq)
    t_init = time.perf_counter() - t0
    dq.norm_A = max(float(blk.abs().sum(1).max()) + 1.0 for blk in dq.F)  # for the interior-point run below
    dq.F = None  # the engine holds its own copy of the blocks
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(100 + (0 if one else rank))
    rnd = lambda k, lo, hi: torch.empty(k, dtype=torch.float64, device="cuda").uniform_(lo, hi, generator=g)
    z, w = rnd(m, 0.1, 1.1), rnd(m, 0.1, 1.1)
    r = [rnd(k, -0.5, 0.5) for k in (n, me, m, m)]
    d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (n, me, m, m)]

    def step():
        mat.factor(None, z, w)
        return mat.solve(None, z, w, *r, *d)

    if world > 1:  # (a C4 step takes about a second: minutes mean a rank is waiting for one that is not coming)
        dog.arm(max(args.watchdog, 60 + 20 * (args.warmup + args.steps)), "warm-up and timed steps")
    for _ in range(args.warmup):
        step()
    kdist.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    kdist.fence()
    elapsed = kdist.max_over_ranks(time.perf_counter() - t0)
    dog.disarm()
    st = mat.stats()
    all_devices = None
    if one:  # which GPU every rank computed on (the bench line shows that N ranks meant N GPUs)
        import torch.distributed as tdist
        all_devices = [None] * world
        tdist.all_gather_object(all_devices, f"{os.uname().nodename}:cuda:{torch.cuda.current_device()}")
    # the same step as a host with its vectors in (pinned) host memory sees it - SURVEY.md 8(d): z, w, r1..r4 go
    # to the device and dx..dw come back per call; extra information, never `value`
    host_rate = None
    if world == 1:
        hin = [t.cpu().pin_memory() for t in [z, w] + r]
        hout = [torch.empty_like(t, device="cpu").pin_memory() for t in d]
        nh = max(1, min(args.steps, 3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nh):
            for src, dst in zip(hin, [z, w] + r):
                dst.copy_(src, non_blocking=True)
            step()
            for src, dst in zip(d, hout):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
        host_rate = nh / (time.perf_counter() - t0)
    # per-kernel-class device time: HIP events on the library's stream around every launch,
    # in a separate untimed pass over the same workload
    mat.set_profile(True)
    nprof = max(2, min(args.steps, 3))
    for _ in range(nprof):
        step()
    prof = mat.profile()
    mat.set_profile(False)
    per_rank, replicas = None, None
    if one:
        # One SCALE run should yield the curve AND its explanation: per rank the device time of the products and of
        # the exchange (its place in the stream to its completion = the wait for the slowest rank + the transfer) ...
        import torch.distributed as tdist
        mine = {"rank": rank, "ms_products": (prof.get("staged_gemm", (0, 0))[0] + prof.get("staged_gemm_upd", (0, 0))[0]) / nprof,
                "ms_exchange_wait": prof.get("exchange", (0, 0))[0] / nprof,
                "ms_control_sized_chain": prof.get("staged_small", (0, 0))[0] / nprof,
                "ms_factor": st["ms_factor"], "ms_solve": st["ms_solve"]}
        # what the exchanges of a factorisation move INTO this rank (every all-gather brings the other ranks' slots), over
        # the time they hold the stream: a LOWER bound of the rate of a link (the time includes the wait for the slowest
        # rank; xGMI is point to point, a gather arrives over world - 1 links at once)
        rx = st["bytes_exchange_factor"] * (world - 1) / world
        mine["gb_received_per_factor"] = rx / 1e9
        mine["gbs_received"] = rx / 1e9 / (mine["ms_exchange_wait"] * 1e-3) if mine["ms_exchange_wait"] > 0 else None
        mine["gbs_per_link"] = mine["gbs_received"] / (world - 1) if mine["gbs_received"] else None
        per_rank = [None] * world
        tdist.all_gather_object(per_rank, mine)
        # ... and, from a second timed pass, the aggregate of N independent systems, one per GPU (the metric's literal
        # "factor+solve/sec @ N GPU": no exchange at all, scaling "weak")
        if not args.no_replicas_pass:
            dog.arm(max(args.watchdog, 240), "replicas pass")
            del mat
            if hasattr(shard, "close"):
                shard.close()
            torch.cuda.empty_cache()
            dq2 = c4_dense(K, nx, nu, seed=rank)
            mat2 = ipmatrix.IpLQDOCP(device=local_rank, device_vectors=True)
            mat2.init_dense(dq2)
            dq2.F = None
            torch.cuda.empty_cache()
            nrep = max(1, min(args.steps, 5))

            def step2():
                mat2.factor(None, z, w)
                return mat2.solve(None, z, w, *r, *d)

            step2()
            kdist.fence()
            t0 = time.perf_counter()
            for _ in range(nrep):
                res2 = step2()
            kdist.fence()
            el2 = kdist.max_over_ranks(time.perf_counter() - t0)
            replicas = {"value": world * nrep / el2, "unit": "KKT factor+solve/s", "steps": nrep, "ms_per_step": 1e3 * el2 / nrep,
                        "scaling": "weak", "residual_rank0": res2,
                        "what": f"{world} independent C4 systems, one per GPU, no collective in the data path"}
            del mat2
            dog.disarm()
    if rank != 0:
        return None
    per_step = {k: v[0] / nprof for k, v in prof.items() if v[1]}
    launches = {k: v[1] / nprof for k, v in prof.items() if v[1]}
    nz, np1 = nx + nu, nx
    # algorithmic work of the recursion per factorisation
    flops_big = K * (2.0 * np1 * np1 * nz + 1.0 * np1 * nz * nz)          # W = V+ F ; G = F'W (lower half)
    if one:  # rank 0's share of the products (its column range; the update products are ~1 % of it)
        flops_big = float(st["flops_local"])
    q = nu
    flops_upd = K * (2.0 * q * q * nx + 1.0 * q * nx * nx)                 # Rm = K^-1 Y ; V = Gxx - Y'Rm (lower half)
    # bytes of the solve's matrix-vector products per stage: V+ twice, F twice, Y and Rm once.  V is symmetric and from
    # 2048 states on only its tiles (64 x 512) on and below the diagonal are read (k_st_symv_tiles): those are the
    # algorithmic bytes then; the rate over the whole matrix is kept beside it for comparison with earlier rounds
    v_full = 1.0 * np1 * np1
    sym = np1 >= 2048 and not os.environ.get("HQPKKT_NO_SYMV")
    v_read = float(sum((bi // 8 + 1) * 64 * 512 for bi in range((np1 + 63) // 64))) if sym else v_full
    bytes_gemv = K * 8.0 * (2.0 * v_read + 2.0 * np1 * nz + 2.0 * q * nx) * (1 + st["refine_rounds"])
    bytes_gemv_full = K * 8.0 * (2.0 * v_full + 2.0 * np1 * nz + 2.0 * q * nx) * (1 + st["refine_rounds"])
    gemm_ms, gemm_launch = per_step.get("staged_gemm", 0.0), launches.get("staged_gemm", 1.0)
    achieved = flops_big / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None
    traffic, traffic_note = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic_c4.json")), reverse=True)  # newest round first
    if pmcs and (nx, nu) == (5000, 50) and not one:
        # HBM bytes per launch of the stream-K dgemm from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
        # workload (separate runs, gfx950 correction applied: profiles/README.md) - only from passes that ran on the
        # kernel sources of this checkout
        seen = []
        for pmc in pmcs:
            rec = json.load(open(pmc))
            if rec.get("_kernel_source_sha16") == kernel_source_sha16():
                traffic = rec.get("k_dgemm_tn_sk", {}).get("hbm_bytes_per_launch")
                traffic_note = "rocprofv3 counter passes: profiles/" + os.path.basename(pmc)
                break
            seen.append(f"{os.path.basename(pmc)}: {rec.get('_kernel_source_sha16')}")
        if traffic is None:
            traffic_note = ("the counter passes under profiles/ were collected on other kernel sources "
                            f"({'; '.join(seen)} != {kernel_source_sha16()}): not quoted")
    elif (nx, nu) == (5000, 50) and not one:
        traffic_note = "no counter passes (profiles/r0*_pmc_traffic_c4.json)"
    roofline = {"kernel": "k_dgemm_tn_sk<lds-dma, 2x4 waves> / k_dgemm_tn<128,128> (W = V+ F, G = F'W)", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS if achieved else None, "traffic": traffic,
                "traffic_note": traffic_note,
                "launches_per_step": gemm_launch, "avg_launch_ms": gemm_ms / gemm_launch if gemm_launch else None,
                "algorithmic_flops_per_launch": flops_big / gemm_launch if gemm_launch else None,
                "algorithmic_flops_per_step": flops_big,
                "algorithmic_bytes_per_launch": 8.0 * (np1 * np1 + 2 * np1 * nz + nz * nz / 2) / 2}
    kernels = {
        "staged_gemm": {"ms_per_step": gemm_ms, "launches_per_step": gemm_launch, "tflops": achieved,
                        "frac_fp64_peak": achieved / FP64_PEAK_TFLOPS if achieved else None},
        "staged_gemm_upd": {"ms_per_step": per_step.get("staged_gemm_upd"), "launches_per_step": launches.get("staged_gemm_upd"),
                            "tflops": flops_upd / (per_step["staged_gemm_upd"] * 1e-3) / 1e12 if per_step.get("staged_gemm_upd") else None},
        "staged_gemv": {"ms_per_step": per_step.get("staged_gemv"), "launches_per_step": launches.get("staged_gemv"),
                        "gbs": bytes_gemv / (per_step["staged_gemv"] * 1e-3) / 1e9 if per_step.get("staged_gemv") else None,
                        "frac_hbm_peak": bytes_gemv / (per_step["staged_gemv"] * 1e-3) / 1e9 / HBM_PEAK_GBS if per_step.get("staged_gemv") else None,
                        "gbs_counting_all_of_v": bytes_gemv_full / (per_step["staged_gemv"] * 1e-3) / 1e9 if per_step.get("staged_gemv") else None,
                        "v_read": "tiles on and below the diagonal" if sym else "whole matrix"},
        "staged_small": {"ms_per_step": per_step.get("staged_small"), "launches_per_step": launches.get("staged_small")},
        "residual": {"ms_per_step": per_step.get("residual"), "launches_per_step": launches.get("residual")},
    }
    fac_ms = sum(per_step.get(k, 0.0) for k in ("staged_gemm", "staged_gemm_upd", "staged_small", "assemble"))
    out = {
        "metric": "KKT factor+solve/sec (fp64)",
        "value": args.steps * (1 if one else world) / elapsed,
        "unit": "KKT factor+solve/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if one else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "vectors": "resident in HBM",
        "value_with_host_vectors": host_rate,  # PCIe per call (33 MB each way in all): what the shim's host sees
        "shard": {"ranks": world, "transport": transport, "comm_ranks": comm_ranks,
                  "devices": sorted(set(all_devices)) if all_devices else None,
                  "bytes_allgather_per_factor": st["bytes_exchange_factor"], "bytes_panels_this_rank": st["bytes_panels"],
                  "flops_rank0": st["flops_local"], "allgathers_per_factor": st["n_exchange_blocks"],
                  "per_rank": per_rank} if one else None,
        "replicas_value": replicas["value"] if replicas else None,
        "replicas": replicas,
        "config": {"workload": f"C4 = BASELINE configs[3], the metric's 10^6-variable DOCP: multistage LQ optimal control QP, K={K} stages, "
                               f"nx={nx} states, nu={nu} controls -> n={n} me={me} m={m}, dense fx/fu handed over as blocks, x_0 fixed, "
                               f"box bounds on u; plugin LQDOCP (STAGED engine), "
                               + (f"ONE system over {world} GPUs (memory sharded: every rank holds its column strip of every F_k and its row strip of every V_k; "
                                  f"per stage the gather of the F blocks, requested a stage ahead, and ONE gather of the blocks of G_xx)"
                                  if one else "one system per GPU"),
                   "stages": K, "nx": nx, "nu": nu, "n": n, "me": me, "m": m, "plugin": "LQDOCP",
                   "kkt_dim_full": n + me + m, "hbm_gb": (st["bytes_panels"] + st["bytes_updates"]) / 1e9},
        "residual": res,
        "refine_rounds": st["refine_rounds"],
        "init_s": t_init,
        "ms_factor": st["ms_factor"], "ms_solve": st["ms_solve"],
        "kernel_ms_per_step": per_step,
        "kernel_launches_per_step": launches,
        "kernels": kernels,
        "factor_model": {"flops_as_implemented": st["flops_factor"], "factor_ms": fac_ms,
                         "tflops_whole_factor": st["flops_factor"] / (st["ms_factor"] * 1e-3) / 1e12 if st["ms_factor"] else None,
                         "frac_fp64_peak_whole_factor": st["flops_factor"] / (st["ms_factor"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS
                         if st["ms_factor"] else None},
        "roofline": roofline,
    }
    if world == 1 and not args.no_ip:
        # second half of BASELINE.json's metric: interior-point iterations per second on the SAME 10^6-variable DOCP,
        # the whole Mehrotra loop device-resident (hqpkkt_mehrotra: per iteration 1 factorisation, 2 solves, the
        # right-hand sides incl. the dense products with the dynamics rows); QP data: c ~ U(-0.5, 0.5), x_0 ~ U(-1, 1), -1 <= u <= 1
        try:
            c4_qp_vectors(dq, nx, nu)
            t0 = time.perf_counter()
            _x, _y, _z, _w, info = mat.mehrotra(dq)
            wall = time.perf_counter() - t0
            # the KKT conditions of what came back, with the dynamics blocks generated once more (the engine holds its own copy)
            Fk = c4_dense(K, nx, nu, seed=0).F
            kkt = c4_kkt_norms(Fk, K, nx, nu, dq.c, dq.b, dq.d, _x, _y, _z, _w)
            del Fk
            out["ip_iterations_c4"] = {"solver": "hqpkkt_mehrotra (device-resident restatement of Hqp_IpsMehrotra), plugin LQDOCP / STAGED",
                                       "iters": info["iters"], "result": info["result"], "factorisations": info["n_factor"],
                                       "solves": info["n_solve"], "seconds": info["ms_total"] * 1e-3, "wall_s": wall,
                                       "ip_iters_per_s": info["iters"] / (info["ms_total"] * 1e-3) if info["ms_total"] else None,
                                       "gap": info["gap"], "kkt": kkt,
                                       "bounds": f"|u| <= {1.2 / (nx / 3.0):.3e}; {100 * kkt['active_fraction'] / 1.0:.1f} % of the 2 K nu bound rows active at the result"}
        except Exception as e:  # never let the secondary measurement break the bench line
            out["ip_iterations_c4"] = {"error": str(e)}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c4(K, nx, nu)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        cb = out["cpu_baseline"]
        if "lqdocp" in cb:  # the same ratio under the other readings of the extrapolation (all of them extrapolations)
            out["speedup_vs_cpu_baseline_readings"] = {
                "Hqp_IpLQDOCP, exponent of the wide samples": out["value"] * cb["lqdocp"]["extrapolated_fit_s"],
                "Hqp_IpLQDOCP, nx^3 from nx=2000": out["value"] * cb["lqdocp"]["extrapolated_cubic_s"],
                "Hqp_IpLQDOCP, exponent of the full-width slice of the build container (3.87)": out["value"] * cb["lqdocp"]["extrapolated_full_width_exponent_s"],
                "Hqp_IpSpBKP, nx^3 from nx=100": out["value"] * cb["spbkp"]["extrapolated_cubic_s"] if "extrapolated_cubic_s" in cb.get("spbkp", {}) else None}
        del mat
        torch.cuda.empty_cache()
        out["staged_small_sizes"] = staged_small_sizes(local_rank)
        # round 1's headline workload (BASELINE configs[1]) through the full-system engine
        a2 = argparse.Namespace(**vars(args))
        # (the GPU has idled through the CPU baseline above - about a minute of host work: enough warm-up steps for its
        # clocks to be back before the 10 timed ones, a step takes 3 ms)
        a2.steps, a2.warmup = min(args.steps, 10), 50
        c2 = bench_c2(a2, extras=False)
        out["c2_banded_kkt"] = {k: c2[k] for k in ("value", "unit", "ms_per_step", "residual", "roofline", "init_s")} if c2 else None
        out["ip_iterations"] = ip_iterations(2000)
        out["mesh_kkt_configs4_standin"] = mesh_kkt_sizes(local_rank)
    return out



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "launchcheck"],
                    help="c4: 10^6-variable multistage DOCP, STAGED engine (the metric's own configuration); "
                         "c2: banded KKT system of dim 10^5, full-system engine")
    ap.add_argument("--stages", type=int, default=200, help="c4: K")
    ap.add_argument("--nx", type=int, default=5000, help="c4: states per stage")
    ap.add_argument("--ctrl", dest="nu", type=int, default=50, help="c4: controls per stage")
    ap.add_argument("--n", type=int, default=40000, help="x variables (C2: 40000)")
    ap.add_argument("--band", type=int, default=80, help="semi-bandwidth of Q / row width of A (C2: 80)")
    ap.add_argument("--mode", default="SpBKP", choices=["SpBKP", "RedSpBKP"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ip", action="store_true", help="c4: skip the interior-point run on the same problem")
    ap.add_argument("--host-vectors", action="store_true",
                    help="z,w,r*,d* as host pointers (the shim's mode): PCIe-inclusive, never the headline value")
    ap.add_argument("--replicas", action="store_true",
                    help="c4, N>1: one independent system per GPU (weak scaling) instead of ONE system over the ranks")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="c4, N>1, one system: libhqpkkt_rccl.so (collectives in the handle's stream) or the "
                         "torch.distributed callback")
    ap.add_argument("--one-system", action="store_true",
                    help="c2, N>1: all ranks factor and solve ONE banded system together (tree engine: subtrees of the "
                         "assembly tree per rank, one all-gather per factor, all-gather + all-reduce per solve; strong "
                         "scaling) instead of one independent system per GPU.  (c4 shards ONE system by default: "
                         "the STAGED engine's column split with one gather of V_k per stage; --replicas for one each)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-gpu: functional check of the N>1 paths on a one-GPU box (not a measurement)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0")
    ap.add_argument("--watchdog", type=int, default=120,
                    help="N>1: seconds a rank may spend in the rendezvous / communicator creation / first collectives "
                         "before it gives up with exit status 86")
    ap.add_argument("--no-replicas-pass", action="store_true",
                    help="c4, N>1, one system: skip the second timed pass (N independent systems, `replicas_value`)")
    ap.add_argument("--no-retry", action="store_true", help="N>1: do not start the ranks a second time with --transport torch")
    ap.add_argument("--leaf-size", type=int, default=0)
    ap.add_argument("--max-pivots", type=int, default=0)
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but this process was started as one of WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")
    return args


def launch_ranks(gpus):
    """`python bench.py --gpus N` outside a launcher: start N FRESH rank processes (one per GPU) with
    torch.distributed.run and hand their output through.  This parent has imported neither torch nor anything
    that touches HIP, and it does not replace itself: the ranks are children, their exit status is ours.
    If the set fails or exceeds its limit (a rank's watchdog fired, a communicator could not be made), ONE fresh
    set is started with --transport torch (torch.distributed's own collectives behind the exchange callback)."""
    import signal
    import socket
    import subprocess

    def run_once(extra, limit_s):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        env["HQPKKT_RUN_NONCE"] = f"{os.getpid()}.{port}"  # part of the name of the RCCL id file, if one is used
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        p = subprocess.Popen(cmd, env=env, cwd=ROOT, start_new_session=True)
        try:
            return p.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            print(f"bench: the {gpus} ranks did not finish within {limit_s} s - ending them", file=sys.stderr, flush=True)
            try:
                os.killpg(p.pid, signal.SIGTERM)  # the process group this parent started, nothing else
                p.wait(timeout=20)
            except Exception:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except Exception:
                    pass
            return 124

    limit = int(os.environ.get("HQPKKT_BENCH_LIMIT_S", "1500"))
    rc = run_once([], limit)
    if rc != 0 and "--no-retry" not in sys.argv and "--transport" not in sys.argv:
        print(f"bench: the ranks ended with status {rc}; one more set with --transport torch", file=sys.stderr, flush=True)
        rc = run_once(["--transport", "torch", "--no-retry"], limit)
    return rc


def bench_c2(args, extras=True):
    import torch

    from hqp_amd import dist as kdist

    rank, local_rank, world = kdist.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dog = Watchdog(rank)
    if world > 1:
        dog.arm(getattr(args, "watchdog", 120), "process group rendezvous")
    if args.share_gpu and world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group(args.backend)
    else:
        kdist.init(args.backend)
    dog.disarm()

    from hqp_amd import ipmatrix, problems

    one = args.one_system and world > 1
    seed_off = 0 if one else rank  # one system: every rank holds the same (replicated) inputs
    prog = problems.banded_qp(args.n, args.band, seed=12345 + seed_off)
    state = problems.ip_state(prog, seed=1 + seed_off)
    cls = ipmatrix.IpSpBKP if args.mode == "SpBKP" else ipmatrix.IpRedSpBKP
    mat = cls(device=local_rank, device_vectors=not args.host_vectors, leaf_size=args.leaf_size,
              max_pivots=args.max_pivots,
              shard=(rank, world, kdist.make_exchange(rank, local_rank)) if one else None)
    t0 = time.perf_counter()
    mat.init(prog)  # analysis (host) + upload; one-time, not part of a step
    t_init = time.perf_counter() - t0
    if args.host_vectors:
        dev = [np.ascontiguousarray(a) for a in state]
        d = [np.zeros(k) for k in (prog.n, prog.me, prog.m, prog.m)]
    else:
        dev = [torch.as_tensor(a).cuda() for a in state]
        d = [torch.zeros(k, dtype=torch.float64, device="cuda") for k in (prog.n, prog.me, prog.m, prog.m)]

    def step():
        mat.factor(prog, dev[0], dev[1])
        return mat.solve(prog, *dev, *d)

    fence = kdist.fence

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = kdist.max_over_ranks(elapsed)
    st = mat.stats()

    # per-kernel-class device time (HIP events on the library's stream around every
    # launch), taken in a separate untimed pass over the same workload
    mat.set_profile(True)
    nprof = max(3, min(args.steps, 10))
    for _ in range(nprof):
        step()
    prof = mat.profile()
    mat.set_profile(False)

    if rank == 0:
        struct = mat.structure()
        work = class_work(struct, 0 if one else None)
        per_step = {k: v[0] / nprof for k, v in prof.items()}
        launches = {k: v[1] / nprof for k, v in prof.items()}
        solves_per_step = 1 + st["refine_rounds"]
        # the fronts whose sweeps run inside k_solve_top (hqp_amd/csrc/solve_top.hip.h: all levels of this tree, in two
        # launches) count under that class, the others under the per-level kernels
        top = mat.debug(31)
        if top[0] > 0 and not one:
            lev = np.asarray(struct["level"])
            p_, b_ = struct["npiv"].astype(np.float64), struct["nborder"].astype(np.float64)
            fused = lev >= top[1]
            sweep = lambda m: {"flops": float((p_[m] ** 2 + 2 * b_[m] * p_[m]).sum()), "bytes": float((8 * (p_[m] ** 2 / 2 + b_[m] * p_[m])).sum())}
            work["solve_fwd"], work["solve_bwd"] = sweep(~fused), sweep(~fused)
            work["solve_top"] = {q: 2 * v for q, v in sweep(fused).items()}
        for k in ("solve_fwd", "solve_bwd", "solve_top"):  # the sweeps run once per solve of the step
            if k in work:
                work[k] = {q: v * solves_per_step for q, v in work[k].items()}
        # every kernel class against both ceilings (algorithmic flops / bytes per step)
        kernels = {k: {"ms_per_step": per_step.get(k, 0.0), "launches_per_step": launches.get(k, 0.0),
                       "tflops": w["flops"] / (per_step[k] * 1e-3) / 1e12 if per_step.get(k) else None,
                       "gbs": w["bytes"] / (per_step[k] * 1e-3) / 1e9 if per_step.get(k) else None,
                       "frac_fp64_peak": w["flops"] / (per_step[k] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if per_step.get(k) else None,
                       "frac_hbm_peak": w["bytes"] / (per_step[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if per_step.get(k) else None}
                   for k, w in work.items()}
        dom = max(("factor_diag", "panel_solve", "schur_update"), key=lambda k: per_step.get(k, 0.0))
        dom_ms = per_step[dom]
        dom_launch_ms = dom_ms / max(launches[dom], 1.0)
        flops_per_launch = work[dom]["flops"] / max(launches[dom], 1.0)
        achieved = flops_per_launch / (dom_launch_ms * 1e-3) / 1e12
        traffic = None  # (no counter passes of the tree engine's kernels since round 1: profiles/pmc_traffic.json is k_factor_diag's)
        roofline = {"kernel": "k_factor_blk" if dom == "factor_diag" else "k_" + dom, "bound": "latency" if dom == "factor_diag" else "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                    "launches_per_step": launches[dom], "avg_launch_ms": dom_launch_ms,
                    "algorithmic_flops_per_step": work[dom]["flops"]}
        # SURVEY.md 8(d) band model of the whole factorisation, for reference
        N, beta = st["dim"], st["sbw"]
        fac_ms = sum(per_step.get(k, 0.0) for k in ("factor_diag", "panel_solve", "schur_update"))
        model = {"flops_band_model": float(N) * beta * beta, "factor_ms": fac_ms,
                 "tflops_band_model": float(N) * beta * beta / (fac_ms * 1e-3) / 1e12 if fac_ms > 0 else None,
                 "tflops_as_implemented": st["flops_factor"] / (fac_ms * 1e-3) / 1e12 if fac_ms > 0 else None}
        out = {
            "metric": "KKT factor+solve/sec (fp64)",
            "value": args.steps * (1 if one else world) / elapsed,
            "unit": "KKT factor+solve/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if one else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "vectors": "host pointers (PCIe per call)" if args.host_vectors else "resident in HBM",
            "config": {"workload": f"C2 synthetic banded KKT: n={prog.n} me={prog.me} m={prog.m} band={args.band} "
                                   f"-> KKT dim {st['dim']}, mat_sbw {st['sbw']}, plugin {args.mode} "
                                   + (f"(ONE system sharded over {world} GPUs)" if one else "(one system per GPU)"),
                       "kkt_dim": st["dim"], "mat_sbw": st["sbw"], "plugin": args.mode,
                       "supernodes": st["n_supernodes"], "tree_levels": st["n_levels"], "max_front": st["max_front"],
                       "nnz_kkt": st["nnz_kkt"], "nnz_factor": st["nnz_factor"]},
            "residual": res,
            "refine_rounds": st["refine_rounds"],
            "n_2x2": st["n_2x2"], "n_perturbed": st["n_perturbed"],
            "init_s": t_init,
            "shard": {"ranks": st["shard_count"], "replicated_top_supernodes": st["n_top"],
                      "exchange_blocks": st["n_exchange_blocks"], "flops_rank0": st["flops_local"],
                      "flops_top": st["flops_top"], "bytes_exchange_factor": st["bytes_exchange_factor"],
                      "bytes_exchange_step": st["bytes_exchange_step"]} if one else None,
            "kernel_ms_per_step": per_step,
            "kernel_launches_per_step": launches,
            "kernels": kernels,
            "factor_model": model,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and extras:
            out["cpu_baseline"] = cpu_baseline(prog, state)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            if not args.host_vectors:
                out["concurrent_systems_per_gpu"] = concurrent_systems(prog, state, cls, local_rank, args.steps)
            out["ip_iterations"] = ip_iterations(2000)
            out["ip_iterations_large"] = ip_iterations(33333)
        return out
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    from hqp_amd import dist as kdist
    if args.workload == "launchcheck":
        # the N > 1 plumbing alone, no GPU needed (tests/test_dist_gloo.py): rendezvous over gloo, the fence and
        # the max-over-ranks of the timing contract; rank 0 prints what it saw
        # (HQPKKT_LAUNCHCHECK=hang: rank 1 never reaches the rendezvous and rank 0's watchdog must end the run;
        # =fail_rccl: the first set of ranks fails unless it was started with --transport torch - the launcher's retry)
        mode = os.environ.get("HQPKKT_LAUNCHCHECK", "")
        rank0, _l, _w = kdist.env_world()
        dog = Watchdog(rank0)
        dog.arm(args.watchdog, "process group rendezvous")
        if mode == "hang" and rank0 == 1:
            time.sleep(3600)
        if mode == "fail_rccl" and args.transport != "torch":
            sys.exit(7)
        rank, _lr, world = kdist.init(backend="gloo")
        dog.disarm()
        kdist.fence(device_sync=False)
        t = kdist.max_over_ranks(float(rank))
        if rank == 0:
            print(json.dumps({"launchcheck": True, "world": world, "max_rank": t, "gpus": args.gpus} |
                             ({"transport": args.transport} if mode else {})))
        kdist.finalize()
        return
    out = bench_c4(args) if args.workload == "c4" else bench_c2(args)
    if out is not None:
        print(json.dumps(out))
    kdist.finalize()


if __name__ == "__main__":
    main()
