"""The drop-in boundary is a C ABI: a plain-C host (tests/c_host/c_abi_host.c, compiled
with gcc against include/hqpkkt.h and linked to libhqpkkt.so) drives analyze / factor /
solve / residual and the device-resident Mehrotra loop without Python, C++ or HIP
headers.  CPU: it builds, and fails loudly (HQPKKT_E_DEVICE) without a GPU.  GPU: the
printed residuals and the IP result are checked."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "c_abi_host")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_host", "c_abi_host.c"), "-o", exe,
           "-L", os.path.join(ROOT, "hqp_amd"), "-lhqpkkt", "-Wl,-rpath," + os.path.join(ROOT, "hqp_amd"), "-lm"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_c_host_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the gpu test")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stdout, out.stderr)
    assert "status 100" in out.stdout  # HQPKKT_E_DEVICE: there is no CPU fallback


@pytest.mark.gpu
def test_c_host_runs_on_the_gpu(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    line = [l for l in out.stdout.splitlines() if l.startswith("C_ABI ok")][-1]
    f = dict(re.findall(r"(\w+) (-?[\d.e+-]+)", line))
    assert int(f["dim"]) == 600 + 300 + 600 and int(f["sbw"]) > 0
    assert float(f["res"]) <= 1e-10 and abs(float(f["res"]) - float(f["res2"])) <= 1e-12
    assert int(f["ip_result"]) == 0 and 1 <= int(f["ip_iters"]) <= 40
    assert float(f["mu"]) <= 1e-9 and float(f["zmin"]) > 0 and float(f["cmin"]) > -1e-8
    # the multistage plugin's dense hand-over (hqpkkt_analyze_staged / hqpkkt_set_values_staged) from plain C
    line = [l for l in out.stdout.splitlines() if l.startswith("C_ABI staged ok")][-1]
    g = dict(re.findall(r"(\w+) (-?[\d.e+-]+)", line))
    assert int(g["stages"]) == 6 and int(g["dim"]) == 6 * 43 + 40 + 6 * 40 + 40
    assert float(g["res"]) <= 1e-10 and abs(float(g["res"]) - float(g["res2"])) <= 1e-12


def test_stage_blocks_from_row_lists_at_the_headline_size(tmp_path):
    """shim/stage_extract.h, the walk that fills LQDOCPHip's dense stage blocks from the row lists of A, on the CPU: a
    small staircase against the blocks it was made from, and BASELINE configs[3]'s 200 stages of 5000 states and 50
    controls (5.05e9 entries in the dynamics rows, generated on the fly, counted by the sink): every count and
    offset on the path is 64-bit (compiled with -Wconversion -Werror; the totals pass 2^32 and come out exactly)."""
    exe = str(tmp_path / "stage_extract_test")
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wconversion", "-Werror=conversion", os.path.join(ROOT, "tests", "c_host", "stage_extract_test.cc"),
           "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout)
    assert "entries 5050000000 (> 2^32), arena elements 5056000000" in out.stdout


def test_detect_stages_matches_the_plan():
    """hqpkkt_detect_stages (three ints per row of A; host-only) finds the stage sizes the STAGED engine's own analysis of
    the CSR form finds: DOCPs with and without controls in a stage, the Prg_DID structure, and what is no staircase."""
    import ctypes as C
    import numpy as np
    from hqp_amd import _lib, problems
    L = _lib.lib()
    ip = C.POINTER(C.c_int)
    L.hqpkkt_detect_stages.argtypes = [C.c_int, C.c_int, ip, ip, ip, C.c_int, ip, ip, ip, ip]

    def detect(prog):
        p, i, _x = prog.A
        p, i = np.asarray(p), np.asarray(i)
        ln = np.diff(p).astype(np.int32)
        last = np.where(ln > 0, i[np.maximum(p[1:] - 1, 0)], -1).astype(np.int32)
        prev = np.where(ln > 1, i[np.maximum(p[1:] - 2, 0)], -1).astype(np.int32)
        K, nd = C.c_int(), C.c_int()
        nx, nu = np.zeros(prog.me + 1, np.int32), np.zeros(prog.me, np.int32)
        e = L.hqpkkt_detect_stages(prog.n, prog.me, ln.ctypes.data_as(ip), last.ctypes.data_as(ip), prev.ctypes.data_as(ip), prog.me,
                                   C.byref(K), nx.ctypes.data_as(ip), nu.ctypes.data_as(ip), C.byref(nd))
        return e, K.value, nx[:K.value + 1].tolist(), nu[:K.value].tolist(), nd.value

    e, K, nx, nu, nd = detect(problems.lq_docp(7, 5, 2, final_eq=2))
    assert (e, K, nx, nu, nd) == (0, 7, [5] * 8, [2] * 7, 35)
    e, K, nx, nu, nd = detect(problems.did_like_qp(50))
    assert e == 0 and K == 50 and nx == [2] * 51 and nu == [1] * 50 and nd == 100
    # two dynamics rows exchanged: the last columns no longer climb -> HQPKKT_E_FORMAT (6)
    prog = problems.lq_docp(4, 3, 1)
    p, i, x = (np.asarray(a).copy() for a in prog.A)
    r0, r1 = slice(p[1], p[2]), slice(p[2], p[3])
    assert p[2] - p[1] == p[3] - p[2]
    i[r0], i[r1] = i[r1].copy(), i[r0].copy()
    assert detect(problems.Program(prog.n, prog.me, prog.m, prog.Q, (p, i, x), prog.C))[0] == 6


def test_host_side_cpp_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md section 5 (sanitizers on the CPU side): analysis.cpp and staged_plan.cpp, the product's host-side C++,
    built with -fsanitize=address,undefined and run over banded, Prg_DID-like and multistage structures (orderings,
    shard plans for 1 / 3 / 8 ranks, big stages, dense hand-over, a non-staircase): no report, exit status 0."""
    exe = str(tmp_path / "sanitize_host")
    src = [os.path.join(ROOT, "tests", "c_host", "sanitize_host.cc"), os.path.join(ROOT, "hqp_amd", "csrc", "analysis.cpp"),
           os.path.join(ROOT, "hqp_amd", "csrc", "staged_plan.cpp")]
    out = subprocess.run(["g++", "-O1", "-g0", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe] + src,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.returncode, run.stdout[-500:], run.stderr[-3000:])
    assert "sanitize_host ok" in run.stdout
