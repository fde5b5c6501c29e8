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
    assert declared == {"hcspmm_dist_create", "hcspmm_dist_destroy", "hcspmm_dist_step", "hcspmm_dist_last_error",
                        "hcspmm_dist_partition_rows", "hcspmm_dist_extract_block"}
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "libhcspmm_dist.so")], stdout=subprocess.PIPE, text=True).stdout
    assert declared <= set(re.findall(r" T (\w+)", nm))
    needed = subprocess.run(["readelf", "-d", os.path.join(CSRC, "libhcspmm.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed
    assert os.path.exists(_build_dist(tmp_path))


def test_dist_host_side_matches_the_python_shard():
    """hcspmm_dist_partition_rows / _extract_block (C ABI) == hcspmm.sharded.partition_rows / ShardedGraph (what the PyTorch
    path uses), on graphs with hubs, empty rows and more ranks than row windows."""
    import ctypes
    import sys
    import numpy as np
    for p in (ROOT, os.path.join(ROOT, "hc-spmm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from hcspmm import graphs
    from hcspmm.sharded import ShardedGraph, partition_rows
    try:
        L = ctypes.CDLL(os.path.join(CSRC, "libhcspmm_dist.so"))
    except OSError as e:  # (needs librccl and the HIP runtime to resolve; present in the ROCm image)
        pytest.skip("libhcspmm_dist.so does not load here: %s" % e)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    L.hcspmm_dist_partition_rows.argtypes = [vp, i64, ctypes.c_int, vp]
    L.hcspmm_dist_extract_block.argtypes = [vp, vp, i64, ctypes.c_int, vp, ctypes.c_int, vp, vp, ctypes.POINTER(i64)]
    cases = [graphs.powerlaw_graph(5003, 90000, seed=11, max_degree_frac=0.3), graphs.uniform_graph(1000, 3000, seed=2),
             graphs.powerlaw_graph(40, 200, seed=3), (np.zeros(101, np.int32), np.zeros(0, np.int32))]
    for rp, col in cases:
        N = len(rp) - 1
        for world in (1, 2, 3, 8):
            ranges = np.zeros(2 * world, np.int64)
            assert L.hcspmm_dist_partition_rows(rp.ctypes.data, N, world, ranges.ctypes.data) == 0
            want = partition_rows(rp, world)
            assert [tuple(r) for r in ranges.reshape(-1, 2).tolist()] == want
            for rank in range(world):
                g = ShardedGraph(rp, col, want, rank)
                rp_out = np.full(g.n_local + 1, -7, np.int32)
                col_out = np.full(max(len(g.column_index), 1), -7, np.int32)
                pad = i64(0)
                assert L.hcspmm_dist_extract_block(rp.ctypes.data, col.ctypes.data if len(col) else None, N, world, ranges.ctypes.data, rank,
                                                   rp_out.ctypes.data, col_out.ctypes.data, ctypes.byref(pad)) == 0
                assert pad.value == g.pad_rows and np.array_equal(rp_out, g.row_pointers)
                assert np.array_equal(col_out[:len(g.column_index)], g.column_index)
    # row pointers that decrease or do not start at 0 are refused before a single column id is read (the interface takes no
    # entry count: rowptr[N] is it)
    rp, col = cases[2]
    for broken in (np.concatenate([rp[:10], rp[10:11] + 50, rp[11:]]).astype(np.int32), (rp + 1).astype(np.int32)):
        ranges = np.zeros(4, np.int64)
        assert L.hcspmm_dist_partition_rows(broken.ctypes.data, 40, 2, ranges.ctypes.data) != 0
        ok = np.array([0, 16, 16, 40], np.int64)
        out_rp, out_col = np.zeros(41, np.int32), np.zeros(len(col) + 64, np.int32)
        assert L.hcspmm_dist_extract_block(broken.ctypes.data, col.ctypes.data, 40, 2, ok.ctypes.data, 0, out_rp.ctypes.data,
                                           out_col.ctypes.data, None) != 0
    bad = np.array([0, 16, 8, 40], np.int64)  # ranges that do not tile the rows
    assert L.hcspmm_dist_extract_block(cases[2][0].ctypes.data, cases[2][1].ctypes.data, 40, 2, bad.ctypes.data, 0, None, None, None) != 0


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
    # the consumer checks the plan counts of its fixed graph: it runs with the library's own plan parameters, whatever stress
    # settings (HCSPMM_SLICE_THRESHOLD, ...) the surrounding test run has in its environment
    env = {k: v for k, v in os.environ.items() if not k.startswith("HCSPMM_")}
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout
    assert "capi_smoke ok" in r.stdout
