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


def _build_dist(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "dist_smoke")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "capi", "dist_smoke.cpp"),
                           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-L", CSRC, "-lhcspmm_dist", "-lhcspmm",
                           "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_dist_library_exports_its_header(tmp_path):
    """include/hcspmm_dist.h <-> libhcspmm_dist.so (the multi-GPU step without PyTorch); libhcspmm.so itself links no RCCL."""
    import re
    header = open(os.path.join(ROOT, "include", "hcspmm_dist.h")).read()
    declared = set(re.findall(r"\b(hcspmm_dist_[a-z_]+)\s*\(", header))
    assert declared == {"hcspmm_dist_create", "hcspmm_dist_destroy", "hcspmm_dist_step", "hcspmm_dist_last_error"}
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "libhcspmm_dist.so")], stdout=subprocess.PIPE, text=True).stdout
    assert declared <= set(re.findall(r" T (\w+)", nm))
    needed = subprocess.run(["readelf", "-d", os.path.join(CSRC, "libhcspmm.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed
    assert os.path.exists(_build_dist(tmp_path))


@pytest.mark.gpu
def test_dist_consumer_runs_through_rccl(tmp_path):
    """One-rank communicator with the collective path forced on a one-GPU box; one process per GPU (up to 4) where more are visible."""
    import torch
    exe = _build_dist(tmp_path)
    world = max(1, min(4, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("dist_smoke ok") == world


def test_capi_consumer_compiles_and_links(tmp_path):
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_capi_consumer_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "capi_smoke ok" in r.stdout
