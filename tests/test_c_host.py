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
