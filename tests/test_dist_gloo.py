"""CPU, world_size 2 over gloo: the N>1 plumbing bench.py uses (unit sharding,
fence, max / sum over ranks) and bench.py's own launcher (`python bench.py --gpus N`
starts its N ranks itself).  The collectives of ONE system that is sharded over the
ranks are covered by tests/test_shard_cpu.py (gloo, numpy model of the kernels) and
tests/test_gpu_shard.py (the kernels themselves, ranks sharing the test box's GPU)."""
import os
import subprocess
import time
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from hqp_amd import dist
    rank, local_rank, world = dist.init(backend="gloo")
    units = dist.shard_units(5, rank, world)
    dist.fence(device_sync=False)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # uneven work
    dist.fence(device_sync=False)
    el = time.perf_counter() - t0
    tmax = dist.max_over_ranks(0.05 * (rank + 1))
    total = dist.sum_over_ranks(len(units))
    if rank == 0:
        print(json.dumps(dict(world=world, units=units, tmax=tmax, total=total, el=el)))
    dist.finalize()
""") % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["units"] == [0, 2, 4]
    assert abs(d["tmax"] - 0.10) < 1e-12          # slowest rank
    assert d["total"] == 5.0                      # every unit owned exactly once
    assert d["el"] >= 0.09                        # rank 0 waited for rank 1 at the fence


def test_shard_units_cover_exactly_once():
    from hqp_amd import dist
    for world in (1, 2, 3, 8):
        owned = sorted(u for r in range(world) for u in dist.shard_units(11, r, world))
        assert owned == list(range(11))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside a launcher (no RANK / WORLD_SIZE in the environment) starts two fresh
    rank processes itself, hands rank 0's line through and returns their exit status."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launchcheck"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"launchcheck": True, "world": 2, "max_rank": 1.0, "gpus": 2}
    # a failing rank's status comes back (here: a world size that contradicts --gpus)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launchcheck"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, RANK="0", WORLD_SIZE="3"))
    assert bad.returncode != 0


def test_bench_watchdog_and_retry():
    """N > 1 must never hang: a rank that waits for one that is not coming gives up after --watchdog seconds
    (exit status 86, the launcher ends the others and returns non-zero); a first set of ranks that fails is
    replaced ONCE by a fresh set with --transport torch."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launchcheck"]
    t0 = time.time()
    hang = subprocess.run(cmd + ["--watchdog", "10", "--no-retry"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                          env=dict(env, HQPKKT_LAUNCHCHECK="hang"))
    assert hang.returncode != 0 and time.time() - t0 < 200
    assert "giving up (exit 86)" in hang.stderr, hang.stderr[-2000:]
    retry = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, HQPKKT_LAUNCHCHECK="fail_rccl"))
    assert retry.returncode == 0, retry.stderr[-2000:]
    assert "one more set with --transport torch" in retry.stderr
    d = json.loads([l for l in retry.stdout.splitlines() if l.startswith("{")][0])
    assert d["world"] == 2 and d["transport"] == "torch"
