"""The C ABI consumed from plain C++ (no Python objects, no torch): tests/capi/capi_smoke.cpp is compiled
with hipcc against include/hcspmm.h + libhcspmm.so and run as its own process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi", "capi_smoke.cpp")
CSRC = os.path.join(ROOT, "hc-spmm_amd", "csrc")


def _build(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "capi_smoke")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", SRC, "-I", os.path.join(ROOT, "include"),
                           "-L", CSRC, "-lhcspmm", "-Wl,-rpath," + CSRC, "-o", exe])
    return exe


def test_capi_consumer_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_capi_consumer_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "capi_smoke ok" in r.stdout
