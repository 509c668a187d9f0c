"""GPU: ONE KKT system sharded over several ranks (hqpkkt_set_shard).
The test box has a single MI355X, so the ranks share cuda:0 and the exchange is
staged through gloo; the kernels, the shard plan, the phase split and the exchange
regions are exactly the multi-GPU ones (with RCCL only the transport differs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _transport(world):
    """Environment of the ranks: with at least `world` GPUs in the box every rank takes its own and the exchange is
    libhqpkkt_rccl.so's (RCCL over xGMI, collectives in the handle's stream); on the one-GPU test box the ranks share
    cuda:0 and the exchange is staged through gloo."""
    import torch
    if torch.cuda.device_count() >= world and world > 1:  # (counting devices does not initialise the GPU)
        return dict(SHARD_BACKEND="nccl", SHARD_TRANSPORT="rccl")
    return dict(SHARD_BACKEND="gloo")


def _run_workers(world, cases, extra_env=None, timeout=900):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free port (concurrent runs must not collide)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SHARD_CASES=json.dumps(cases), MASTER_ADDR="127.0.0.1", **_transport(world))
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    return json.loads(line[len("SHARD_RESULT "):])

CASES = [["banded", 1500, 12, "SpBKP"], ["banded", 1500, 12, "RedSpBKP"], ["docp", 24, 6, 3, "SpBKP"],
         ["did", 400, "RedSpBKP"], ["grid", 40, 40, 1, "RedSpBKP"]]


@pytest.mark.parametrize("world,port", [(2, 29561), (3, 29562), (4, 29563)])
def test_sharded_system_matches_single(world, port):
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps(CASES), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    per_rank = json.loads(line[len("SHARD_RESULT "):])
    assert len(per_rank) == world
    for ci, case in enumerate(CASES):
        recs = [r[ci] for r in per_rank]
        r0 = recs[0]
        # the sharded factorisation is the same elimination: same solution up to the
        # refinement target, and the reference's residual bound holds on every rank
        assert r0["diff"] < 1e-9, (case, r0)
        for r in recs:
            assert r["res"] <= 1e-10 and r["same_as_rank0"], (case, r)
            assert r["top"] == r0["top"] == r["n_top"] and r["top"] > 0
        # every supernode is either replicated or owned by exactly one rank
        assert sum(r["owned"] for r in recs) + r0["top"] == r0["nodes"], (case, recs)
        assert r0["xblocks"] >= world - 1


def test_sharded_system_switches_the_zero_diagonal_placement_on_every_rank():
    """zd_policy -1 on weak Hessian diagonals (Prg_DID, qx = 1e-4) with w/z over 12 decades: the refinement of the first
    solve fails and the handle re-analyses with the other placement (hqpkkt_opts.zd_policy) - sharded too: every rank
    sees the same residual, switches and factorises again; same solution as the unsharded handle."""
    cases = [["did_spread", 200, 1, 6.0, "RedSpBKP"]]
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps(cases), MASTER_ADDR="127.0.0.1", HQPKKT_TRACE_SOLVE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29567", os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert out.stderr.count("zero-diagonal placement 2 -> 0") >= 3  # both ranks and the unsharded partner
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    per_rank = json.loads(line[len("SHARD_RESULT "):])
    r0 = per_rank[0][0]
    assert r0["diff"] < 1e-7, r0
    for r in per_rank:
        assert r[0]["same_as_rank0"] and r[0]["res"] <= 10 * r0["res_single"] + 1e-10, r[0]


# (the third: 200 controls per stage - the blocked elimination of K on the second stream, beside the strips' products)
STAGED_CASES = [["docp", 5, 300, 6, "LQDOCP"], ["docp", 3, 520, 20, "LQDOCP"], ["docp", 3, 400, 200, "LQDOCP"]]


@pytest.mark.parametrize("world,port", [(2, 29571), (3, 29572)])
def test_sharded_staged_system_matches_single(world, port):
    """STAGED engine over several ranks: the state columns of a stage's three products cut into one
    range per rank, ONE all-gather per stage (the strips of V_k).  Same solution as the unsharded
    handle, identical vectors on all ranks."""
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps(STAGED_CASES), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    per_rank = json.loads(line[len("SHARD_RESULT "):])
    for ci, case in enumerate(STAGED_CASES):
        recs = [r[ci] for r in per_rank]
        assert recs[0]["diff"] < 1e-9, (case, recs[0])
        cuts = recs[0]["cuts"]
        assert cuts[0] == 0 and cuts[-1] == case[2] and all(a <= b for a, b in zip(cuts, cuts[1:]))
        assert all(c % 128 == 0 or c == case[2] for c in cuts) and sum(1 for a, b in zip(cuts, cuts[1:]) if b > a) >= 2
        for r in recs:
            assert r["res"] <= 1e-10 and r["same_as_rank0"] and r["ranks"] == world, (case, r)


# the stage shapes of the recursion: final-state rows carried back through the stages (the carried rows' columns travel
# with the first exchange), path equalities that consume controls, a FREE initial state (v_0 gathered), state bounds,
# w/z spread over decades; the last with more ranks than 128-column blocks (empty strips)
STAGED_SHAPES = [["docpx", 6, 300, 5, dict(x0_fixed=True, final_eq=7), 1.0, "LQDOCP"],
                 ["docpx", 5, 280, 6, dict(x0_fixed=False, final_eq=4, path_eq=2), 2.0, "LQDOCP"],
                 ["docpx", 4, 520, 12, dict(x0_fixed=False, path_eq=3, path_eq_every=2, x_bounds=40), 1.0, "LQDOCP"],
                 ["docpx", 7, 140, 4, dict(x0_fixed=True, final_eq=3, x_bounds=10), 3.0, "LQDOCP"]]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_memory_sharded_stages_on_every_stage_shape(world):
    """the memory-sharded partition (every rank holds its column strip of F_k and its row strip of
    V_k only) with 2, 3 and 4 ranks against the unsharded handle on the same system: solutions equal to 1e-9, the
    residual bound on every rank, identical vectors on all ranks.  Two ranks: the pair of blocks half the ring apart is
    cut in two; three: every pair has one owner; four: both kinds."""
    per_rank = _run_workers(world, STAGED_SHAPES)
    for ci, case in enumerate(STAGED_SHAPES):
        recs = [r[ci] for r in per_rank]
        assert recs[0]["diff"] < 1e-9, (case, recs[0])
        for r in recs:
            assert r["res"] <= 1e-10 and r["same_as_rank0"] and r["ranks"] == world, (case, r)


@pytest.mark.parametrize("world", [2, 3])
def test_memory_sharded_stages_on_random_multistage_qps(world):
    """60 cases of tools/fuzz_staged.py (random stages, states, controls, fixed / free x_0, carried final-state rows, path
    equalities, state bounds, w/z spreads) through the sharded handle and the unsharded one: the same status on every
    rank, and where both solve, the same solution to 1e-8 (stages with an odd number of states are refused by the sharded
    plan: E_SIZES on every rank)."""
    recs = [r[0] for r in _run_workers(world, [["fuzzst", 0, 60, "LQDOCP"]])]
    assert all(r["same_status"] for r in recs), recs
    assert recs[0]["worst"] <= 1e-8 and recs[0]["compared"] >= 15, recs[0]


@pytest.mark.parametrize("world", [3, 4])
def test_memory_sharded_dense_hand_over(world):
    """The dense hand-over (hqpkkt_set_values_staged: every rank copies its columns of the caller's blocks) and the
    products of residuum() with the local blocks (summed over the ranks), stages of 1024 states: same solution as the
    unsharded handle, and the rank's arenas hold about 1 / P of the unsharded ones (strips are whole 128-column blocks:
    8 blocks over 3 ranks = 3 + 3 + 2, so 12.5 % more than a third here; the 10 % bound is checked at full stage width)."""
    case = ["c4dense", 12, 1024, 16, "LQDOCP"]
    recs = [r[0] for r in _run_workers(world, [case])]
    assert recs[0]["diff"] < 1e-9, recs[0]
    for r in recs:
        assert r["res"] <= 1e-10 and r["same_as_rank0"] and r["ranks"] == world, r
        assert r["bytes_panels"] <= recs[0]["bytes_panels_single"] * (1.0 / world) * 1.20, (r["bytes_panels"], recs[0]["bytes_panels_single"])


def test_sharded_staged_system_at_full_stage_width():
    """configs[3]'s stage width (nx = 5000, nu = 50, K = 20 stages, dense hand-over) over two ranks that share
    the one GPU of the test box: the stream-K column slices, the pack / unpack of the strips of V_k and the
    exact-size broadcast sequence at the size the multi-GPU bench runs them.  Same solution as the unsharded
    handle to 1e-9, identical vectors on both ranks, no refinement round needed."""
    case = ["c4dense", 20, 5000, 50, "LQDOCP"]
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps([case]), MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free port (concurrent runs must not collide)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    recs = [r[0] for r in json.loads(line[len("SHARD_RESULT "):])]
    assert recs[0]["diff"] < 1e-9, recs[0]
    cuts = recs[0]["cuts"]
    assert cuts[0] == 0 and cuts[-1] == 5000 and 0 < cuts[1] < 5000 and cuts[1] % 128 == 0
    for r in recs:
        assert r["res"] <= 1e-10 and r["same_as_rank0"] and r["ranks"] == 2, r
    # both ranks do about half of the products
    f0, f1 = recs[0]["flops_local"], recs[1]["flops_local"]
    assert abs(f0 - f1) <= 0.1 * max(f0, f1), (f0, f1)
    # ... and hold about half of the stage blocks (<= 1 / P + 10 %)
    for r in recs:
        assert r["bytes_panels"] <= recs[0]["bytes_panels_single"] * 0.5 * 1.10, (r["bytes_panels"], recs[0]["bytes_panels_single"])
    # ... as the allocator sees it: the handles of BOTH ranks together (they share the test box's one GPU) take what the
    # one unsharded handle takes for its stage blocks, plus their work blocks (two whole V, two gathered-F buffers, W, G,
    # exchange slots each: 1.3 GB per rank at this width, whatever the number of stages)
    both, single = recs[0]["hbm_all_ranks"], recs[0]["hbm_single"]
    assert both <= single + 2 * 1.6e9, (both, single)


def test_sharded_staged_system_with_an_odd_number_of_states_is_refused():
    """with an odd number of states per stage the control columns F + nn of the dynamics would start 8, not
    16 bytes aligned under the column split.  The plan of a SHARDED system refuses such a stage (staged_plan.cpp: status
    E_SIZES, on every rank alike - nobody is left in a collective); the unsharded engine takes it (operands that are
    not 16-byte aligned are staged through registers, st_gemm)."""
    import socket
    sys.path.insert(0, ROOT)
    import bench
    from hqp_amd import ipmatrix
    mat = ipmatrix.IpLQDOCP(device_vectors=True)
    mat.init_dense(bench.c4_dense(3, 1001, 40, seed=1))  # unsharded: accepted
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    case = ["c4dense", 3, 1001, 40, "LQDOCP"]
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps([case]), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert out.stderr.count("hqpkkt status 1 in init_dense") == 2, out.stderr[-3000:]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside a launcher starts its ranks itself (fresh child processes) and prints
    ONE strong-scaling line; here two ranks on the one GPU with the exchange staged through gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--stages", "6",
           "--nx", "1024", "--steps", "2", "--warmup", "1", "--no-ip", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["residual"] <= 1e-10
    assert rec["shard"]["ranks"] == 2 and rec["shard"]["comm_ranks"] == 2 and "gloo" in rec["shard"]["transport"]


def test_rccl_transport_single_rank():
    """libhqpkkt_rccl.so on the one GPU of the test box: a communicator of one rank, the STAGED
    engine's exchange path with the collectives in the handle's stream (hqpkkt_set_shard_stream).
    The several-rank form of the same code runs in bench.py --one-system."""
    cases = [STAGED_CASES[0], STAGED_CASES[2]]
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_TRANSPORT="rccl", SHARD_CASES=json.dumps(cases), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29575", os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    for rec in json.loads(line[len("SHARD_RESULT "):])[0]:
        assert rec["diff"] < 1e-9 and rec["res"] <= 1e-10, rec


@pytest.mark.parametrize("world,port", [(2, 29577), (3, 29578)])
def test_sharded_singular_system_same_status_on_all_ranks(world, port):
    """A rank-deficient equality block: the exactly zero pivot appears in one rank's subtree only; the
    status words are agreed by an all-reduce, so every rank raises E_SING (none hangs in the next
    collective)."""
    # (the full plugin is left with a multiplier pivot of ~1e-17 instead of an exact zero: the "tiny pivot"
    # mark, flags[5], set on the owner of that subtree only - it travels with the agreed status words too)
    cases = [["singular", 1500, 12, "RedSpBKP"], ["banded", 1500, 12, "RedSpBKP"], ["singular", 1500, 12, "SpBKP"],
             ["banded", 1500, 12, "SpBKP"]]
    env = dict(os.environ, SHARD_BACKEND="gloo", SHARD_CASES=json.dumps(cases), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("SHARD_RESULT ")][-1]
    per_rank = json.loads(line[len("SHARD_RESULT "):])
    for ci in (0, 2):
        codes = [r[ci]["code"] for r in per_rank]
        assert codes == [4] * world, (cases[ci], codes)
    # ... and the ranks go on together: the next (regular) system is solved as usual
    for r in per_rank:
        for ci in (1, 3):
            assert r[ci]["res"] <= 1e-10 and r[ci]["same_as_rank0"]
