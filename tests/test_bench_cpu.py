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


def _run_bench(argv, env=None, timeout=180):
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_plain_spelling_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns the two ranks itself and relays rank 0's single line
    (HCSPMM_BENCH_DRYRUN=1: the ranks rendezvous over gloo and stop before anything needs a GPU)."""
    r = _run_bench(["--gpus", "2", "--steps", "3"], {"HCSPMM_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["world_size"] == 2 and out["sum_of_ranks_plus_one"] == 3.0 and out["launcher"] == "self" and out["backend"] == "gloo"


def test_more_ranks_than_gpus_is_an_error_that_names_the_count():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs fewer than 2 visible GPUs")
    r = _run_bench(["--gpus", "2", "--steps", "3"])
    assert r.returncode == 3 and r.stdout == ""
    assert "needs 2 visible GPUs, found %d" % torch.cuda.device_count() in r.stderr


def test_a_failing_rank_fails_the_job():
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box the rehearsal succeeds (tests/test_sharded_cpu.py runs it there)")
    r = _run_bench(["--gpus", "2", "--steps", "2", "--workload", "cora"], {"HCSPMM_BENCH_REHEARSAL": "1"})
    assert r.returncode != 0 and "no GPU visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_strong_scaling_blocks_are_the_partition_of_one_graph():
    """--workload products / powerlaw16m / c5: every world size sees the same graph, and a rank's block is the block
    hcspmm.sharded.partition_rows (nnz-balanced, window-aligned) cuts for it."""
    import numpy as np
    from hcspmm.sharded import partition_rows
    for wl in ("products", "c5"):
        full_rp, full_col, n1, n_total = bench.make_strong_block(wl, 1, 0, scale=64)
        assert n1 == n_total == len(full_rp) - 1 and n_total % (16 * bench.STRONG_CHUNKS) == 0
        for world in (2, 8):
            blocks = [bench.make_strong_block(wl, world, r, scale=64) for r in range(world)]
            ranges = [(r * n_total // world, (r + 1) * n_total // world) for r in range(world)]
            # chunks have equal heights and (to a fraction of a percent) equal entry counts: the nnz-balanced cuts of
            # partition_rows fall on the chunk boundaries, give or take a row window or two
            for (a0, a1), (b0, b1) in zip(partition_rows(full_rp, world), ranges):
                assert abs(a0 - b0) <= max(32, n_total // world // 200) and abs(a1 - b1) <= max(32, n_total // world // 200)
            for r, (rp, col, n_local, nt) in enumerate(blocks):
                r0, r1 = ranges[r]
                assert nt == n_total and n_local == r1 - r0
                assert np.array_equal(rp.astype(np.int64), full_rp[r0:r1 + 1].astype(np.int64) - int(full_rp[r0]))
                assert np.array_equal(col, full_col[full_rp[r0]:full_rp[r1]])
    with pytest.raises(SystemExit):
        bench.make_strong_block("products", 3, 0, scale=64)


def test_counter_segments_follow_the_preprocess_marker(tmp_path):
    """One rocprofv3 pass covers every case: the counter rows are cut at each case's edge_to_row_kernel dispatch."""
    d = tmp_path / "pmc_fetch" / "host"
    d.mkdir(parents=True)
    hdr = "Correlation_Id,Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    rows = []

    def add(did, name, val):
        rows.append('%d,%d,"%s","FETCH_SIZE",%f\n' % (did, did, name, val))
    e2r, hyb, fix = "(anonymous namespace)::edge_to_row_kernel(int const*, int, long long, int*)", \
        "void hcspmm::hybrid_plan_kernel<hcspmm::F32, 8, 4, 8, 5, false>(hcspmm::PlanArgs)", "void hcspmm::fixup_kernel<hcspmm::F32, 4>(hcspmm::PlanArgs)"
    add(1, e2r, 5.0); add(2, "copyBuffer", 1.0); add(3, hyb, 100.0); add(4, fix, 10.0); add(5, hyb, 300.0); add(6, fix, 30.0)
    add(7, e2r, 5.0); add(8, hyb, 7.0)
    (d / "1_counter_collection.csv").write_text(hdr + "".join(reversed(rows)))  # order in the file does not matter
    segs = bench._read_counter_segments(str(tmp_path / "pmc_fetch"))
    assert len(segs) == 2
    assert segs[0]["FETCH_SIZE"] == {"void hcspmm::hybrid_plan_kernel<hcspmm::F32, 8, 4, 8, 5, false>": 200.0, "void hcspmm::fixup_kernel<hcspmm::F32, 4>": 20.0}
    assert segs[1]["FETCH_SIZE"] == {"void hcspmm::hybrid_plan_kernel<hcspmm::F32, 8, 4, 8, 5, false>": 7.0}


def test_loi_block_plan_and_case_keys():
    """The `loi` block's cases: {as shuffled, after the reorder} x {the reference's classifier, the refit for the width} at both
    widths, plus the reordered graph with every window on the sparse-row path; every case has its own counter key."""
    plan = bench.loi_plan()
    assert len(plan) == 10 and len({bench.case_key(w, D, "f32", r) for w, D, r in plan}) == 10
    for D in (128, 32):
        for w in ("community", "community_loi"):
            assert (w, D, 0) in plan and (w, D, bench.mi355x_rule(D)) in plan
        assert ("community_loi", D, 2) in plan and ("community", D, 2) not in plan
    assert bench.mi355x_rule(32) == 3 and bench.mi355x_rule(64) == 4 and bench.DEFAULT_RULE == 3
    assert bench.case_key("reddit", 128) == "reddit_d128" and bench.case_key("community_loi", 32, "f32", 3) == "community_loi_d32_r3"
    assert bench.case_key("reddit", 128, "bf16") == "reddit_d128_bf16"


def test_loi_block_host_side_and_oracle_sample_check():
    """reorder_for_loi_block on a small community graph (the relaxed reorder, the permutation applied, the exact reorder's time) and
    the sampled-row oracle check the block runs on every case -- here against a product made by the oracle itself, and against a
    corrupted one."""
    import numpy as np
    import torch
    import oracle
    from hcspmm import graphs
    rp, col, grp = graphs.community_graph(30000, 63000, seed=2)
    info = {}
    rpr, colr = bench.reorder_for_loi_block(rp, col, info)
    assert len(rpr) == len(rp) and len(colr) == len(col) and int(rpr[-1]) == len(col)
    assert info["reorder_s"] > 0 and info["apply_permutation_s"] > 0 and info["full_groups"] > 0 and info["exact"]["full_groups"] > 0
    assert sorted(np.diff(rpr).tolist()) == sorted(np.diff(rp).tolist())  # an isomorphic graph: the same degree multiset
    X = np.random.default_rng(0).standard_normal((30000, 16)).astype(np.float32)
    Z = oracle.spmm_f32(rpr, colr, X)
    ok = bench.oracle_sample_check(rpr, colr, torch.from_numpy(X), torch.from_numpy(Z), n_rows=500)
    assert ok["ok"] and ok["rows"] >= 500 and ok["worst_over_1e-5_bar"] <= 1.0
    Zbad = Z.copy()
    Zbad[np.argsort(np.diff(rpr))[-1]] += 1.0  # the longest row is always in the sample
    assert not bench.oracle_sample_check(rpr, colr, torch.from_numpy(X), torch.from_numpy(Zbad), n_rows=500)["ok"]
