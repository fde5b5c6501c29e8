"""CPU checks of bench.py's bookkeeping (no GPU): the byte formulas of SURVEY.md 8(d), the roofline block -- `frac` never
above 1 and always labelled with its basis -- and the recorded-profile table it falls back to."""
import json
import os
import types

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(N=233000, E=11600000, D=128, kernel_ms=0.47, x_rows=None, n_dense=0, nnz_dense=0, uniq_dense=0, k_sum=0, cols_ref=None):
    h = types.SimpleNamespace(nnz_dense=nnz_dense, uniq_dense=uniq_dense, n_dense=n_dense, dense_k_sum=k_sum)
    return {"header": h, "D": D, "N": N, "E": E, "elem": 4, "kernel_ms": kernel_ms, "x_rows": x_rows or N,
            "cols_referenced": cols_ref or N}


def test_byte_formulas_match_survey_8d():
    c = _case()
    # 4*E*D + 4*N*D + 4*E + 4*(N+1)
    assert bench.algorithmic_bytes(c["N"], c["E"], c["D"], c["header"]) == 4 * 11600000 * 128 + 4 * 233000 * 128 + 4 * 11600000 + 4 * 233001
    # dense windows: 4*uniq*D instead of 4*nnz*D, plus 8*nnz (edgeToColumn + edgeToRow)
    d = _case(n_dense=10, nnz_dense=1000, uniq_dense=200)
    assert bench.algorithmic_bytes(d["N"], d["E"], d["D"], d["header"]) == \
        4 * ((11600000 - 1000) * 128 + 200 * 128 + 233000 * 128) + 8 * 1000 + 4 * 11600000 + 4 * 233001
    assert bench.compulsory_bytes(233000, 11600000, 128, 233000) == 8 * 233000 * 128 + 4 * 11600000 + 4 * 233001


def test_roofline_fraction_is_a_fraction():
    c = _case()
    # X (119 MB) fits the Infinity Cache: algorithmic bytes / t exceeds the HBM peak, so frac must come from elsewhere
    r = bench.roofline_of(c, traffic=3.36e9, traffic_source="test")
    assert r["frac_algorithmic"] > 1.0 and r["bound"] == "l2-miss / Infinity Cache"
    assert 0.85 < r["frac"] < 0.95 and "measured" in r["frac_basis"] and r["traffic_source"] == "test"
    assert abs(r["traffic_over_compulsory"] - 3.36e9 / r["compulsory_bytes"]) < 1e-9
    # no counters available: the compulsory fraction, never the > 1 algorithmic one
    r = bench.roofline_of(c)
    assert r["traffic"] is None and r["frac"] == r["frac_hbm_compulsory"] < 0.1 and "compulsory" in r["frac_basis"]
    # X beyond the Infinity Cache and algorithmic rate below the peak: the contract's own definition
    big = _case(N=2000000, E=32000000, D=128, kernel_ms=2.5, x_rows=16000000, cols_ref=5000000)
    r = bench.roofline_of(big)
    assert r["bound"] == "hbm" and r["frac"] == r["frac_algorithmic"] <= 1.0 and "algorithmic" in r["frac_basis"]
    # dense-tile path: exactly 2*16*K*D flop per window
    dn = _case(n_dense=100, nnz_dense=5000, uniq_dense=2000, k_sum=2400)
    r = bench.roofline_of(dn, traffic=1e9, traffic_source="t")
    assert r["dense_path"]["flops_per_launch"] == 2 * 16 * 2400 * 128


def test_recorded_profiles_carry_their_provenance():
    path = os.path.join(ROOT, "profiles", "measured.json")
    if not os.path.exists(path):
        pytest.skip("no profiles/measured.json in this tree")
    table = json.load(open(path))
    assert "reddit_d128" in table and "c5_share_d128" in table
    for key, e in table.items():
        assert e["traffic_bytes"] > 0 and e["kernel_ms"] > 0 and e["kernel_src_sha"] and os.path.exists(os.path.join(ROOT, e["source"])), key
        rec = bench.recorded_profile(key)
        assert rec["stale"] == (e["kernel_src_sha"] != bench.kernel_src_sha())


def test_every_workload_has_a_generator_entry():
    for name, (n, e, d, vw, desc) in bench.WORKLOADS.items():
        assert n > 0 and d > 0 and vw >= 1 and desc
    assert bench.WORKLOADS["reddit"][:3] == (233000, 11600000, 128)   # BASELINE config 3: the headline
    assert bench.WORKLOADS["c5_share"][:4] == (2000000, 32000000, 128, 8)  # BASELINE config 5, one GPU's share of 16 M / 256 M
