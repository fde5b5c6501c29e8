"""CPU tests that pin the ORACLE (oracle/): against the known-answer table of SURVEY.md Appendix A,
hand-derived cases, scipy, and the golden vectors produced by the reference's own LOI.cpp and
dataset.py (tests/golden/, generators committed next to them)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from hcspmm import graphs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# (uniq, nnz_w) -> (blockPartition, logit, intended type)
KAT = [
    (1, 1, 1, -3.200620, 1), (8, 64, 1, -5.048468, 1), (16, 16, 2, -0.582253, 1), (24, 48, 3, +0.594942, 0),
    (24, 126, 3, -0.741223, 1), (25, 126, 4, -0.003078, 1), (32, 100, 4, +1.720745, 0), (40, 48, 5, +4.100488, 0),
    (45, 700, 6, -0.409070, 1), (130, 130, 17, +22.069473, 0),
]


@pytest.mark.parametrize("uniq,nnz,bp,logit,typ", KAT)
def test_classifier_known_answers(oracle_mod, uniq, nnz, bp, logit, typ):
    size = uniq - 1
    num = (size + 8) // 8
    assert num == bp
    assert abs(oracle_mod.logit(size, nnz, num) - logit) < 5e-7
    assert oracle_mod.classify(size, nnz, num, oracle_mod.RULE_INTENDED) == typ
    assert oracle_mod.classify(size, nnz, num, oracle_mod.RULE_AS_SHIPPED) == 0  # float-as-bool: always 0
    guard = 0 if (size > 32 or logit > 0) else 1
    assert oracle_mod.classify(size, nnz, num, oracle_mod.RULE_INTENDED_GUARD) == guard


def test_classifier_float_division_is_single_precision(oracle_mod):
    # (float)nnz / (int) is a FLOAT division in the reference expression (hybrid_all_kernel.cu:262)
    size, nnz, num = 20, 77, 3
    dens32 = np.float32(nnz) / np.float32(num * 128)
    want = float(np.float32(size)) * 0.19854024 - float(dens32) * 6.578043 - 3.14922857
    assert oracle_mod.logit(size, nnz, num) == want


def test_preprocess_hand_case(oracle_mod):
    # 18 nodes -> 2 windows.  window 0: rows 0..15, row 0 -> {3, 7}, row 2 -> {3, 17}, row 15 -> {0};
    # window 1: row 16 -> {5}, row 17 empty.
    deg = np.zeros(18, np.int64)
    deg[0], deg[2], deg[15], deg[16] = 2, 2, 1, 1
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.array([3, 7, 3, 17, 0, 5], np.int32)
    bp, e2c, e2r, ht = oracle_mod.preprocess(rowptr, col, oracle_mod.RULE_INTENDED)
    # window 0 unique cols sorted: [0, 3, 7, 17] -> size 3, 1 block; ranks: 3->1, 7->2, 17->3, 0->0
    assert bp.tolist() == [1, 1]
    assert e2c.tolist() == [1, 2, 1, 3, 0, 0]
    assert e2r.tolist() == [0, 0, 2, 2, 15, 16]
    assert ht.tolist() == [1, 1]  # tiny windows -> dense-tile path under the intended rule
    assert oracle_mod.preprocess(rowptr, col, oracle_mod.RULE_AS_SHIPPED)[3].tolist() == [0, 0]


def test_preprocess_empty_windows_defined_zero(oracle_mod):
    rowptr = np.zeros(40 + 1, np.int32)
    rowptr[33:] = 2  # only row 32 (window 2) has entries
    col = np.array([1, 9], np.int32)
    bp, e2c, e2r, ht = oracle_mod.preprocess(rowptr, col)
    assert bp.tolist() == [0, 0, 1] and ht.tolist() == [0, 0, 1]
    assert e2c.tolist() == [0, 1] and e2r.tolist() == [32, 32]


def test_preprocess_matches_numpy_definition(oracle_mod):
    rp, col = graphs.powerlaw_graph(1000, 9000, seed=11)
    bp, e2c, e2r, ht = oracle_mod.preprocess(rp, col)
    N = len(rp) - 1
    for w in range((N + 15) // 16):
        lo, hi = rp[w * 16], rp[min(w * 16 + 16, N)]
        if hi == lo:
            assert bp[w] == 0 and ht[w] == 0
            continue
        U = np.unique(col[lo:hi])
        assert bp[w] == (len(U) + 7) // 8
        assert np.array_equal(e2c[lo:hi], np.searchsorted(U, col[lo:hi]))
    assert np.array_equal(e2r, np.repeat(np.arange(N), np.diff(rp)))


@pytest.mark.parametrize("D", [1, 16, 22, 32, 96])
def test_spmm_oracle_against_scipy(oracle_mod, D):
    rp, col = graphs.powerlaw_graph(700, 6000, seed=4)
    N = len(rp) - 1
    X = np.random.default_rng(D).standard_normal((N, D)).astype(np.float32)
    A = sp.csr_matrix((np.ones(len(col), np.float64), col, rp), shape=(N, N))
    z64 = oracle_mod.spmm_f64(rp, col, X)
    assert np.allclose(z64, A @ X.astype(np.float64), rtol=1e-12, atol=1e-12)
    z32 = oracle_mod.spmm_f32(rp, col, X)
    ok, ratio = oracle_mod.check_spmm(z32, rp, col, X)
    assert ok, ratio
    # integer-valued X (GNN_model.py:13-23 idea): exact in fp32, any summation order
    Xi = np.tile(np.arange(N, dtype=np.float32)[:, None], (1, D))
    zi = oracle_mod.spmm_f32(rp, col, Xi)
    assert np.array_equal(zi, (A @ Xi.astype(np.float64)).astype(np.float32))


@pytest.mark.parametrize("rule", [0, 1, 2])
def test_hybrid_dataflow_oracle_is_bit_identical(oracle_mod, rule):
    # consuming edgeToColumn / edgeToRow / blockPartition the way the reference kernel does must give
    # the same bits as the CSR-order sum (ascending unique columns == CSR order within a row)
    for gen, seed in ((graphs.powerlaw_graph, 2), (graphs.planted_dense_graph, 3)):
        rp, col = gen(523, 4000, seed=seed) if gen is graphs.powerlaw_graph else gen(523, seed=seed)
        N = len(rp) - 1
        X = np.random.default_rng(0).standard_normal((N, 24)).astype(np.float32)
        bp, e2c, e2r, ht = oracle_mod.preprocess(rp, col, rule)
        zh = oracle_mod.spmm_hybrid_f32(rp, col, bp, e2c, e2r, ht, X)
        assert np.array_equal(zh, oracle_mod.spmm_f32(rp, col, X))
        # and with every window forced onto the dense-tile data flow
        zh1 = oracle_mod.spmm_hybrid_f32(rp, col, bp, e2c, e2r, np.ones_like(ht), X)
        assert np.array_equal(zh1, oracle_mod.spmm_f32(rp, col, X))


def test_fused_oracle(oracle_mod):
    rp, col = graphs.powerlaw_graph(300, 2000, seed=8)
    N = len(rp) - 1
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, 32)).astype(np.float32)
    Wt = rng.standard_normal((32, 22)).astype(np.float32)
    out, out2 = oracle_mod.spmm_fused_f32(rp, col, X, Wt)
    assert np.array_equal(out2, oracle_mod.spmm_f32(rp, col, X))
    assert np.allclose(out, out2.astype(np.float64) @ Wt.astype(np.float64), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "loi_*.npz")) if "loi_win_" not in p))
def test_loi_oracle_matches_reference_golden(path):
    """oracle/loi_oracle.py vs the permutation the compiled reference LOI.cpp produced."""
    from oracle import loi_oracle
    g = np.load(path)
    rp, col = g["row_pointers"], g["column_index"]
    groups, visit = loi_oracle.reorder_new_direct(rp, col, len(rp) - 1)
    assert [len(x) for x in groups] == g["group_sizes"].tolist()
    assert np.array_equal(np.concatenate([np.asarray(x, np.int32) for x in groups]), g["group_members"])
    assert np.array_equal(loi_oracle.final_order(groups, visit), g["order"])
    groups, visit = loi_oracle.reorder_new(rp, col, len(rp) - 1)  # the other un-windowed variant
    assert [len(x) for x in groups] == g["group_sizes_new"].tolist()
    assert np.array_equal(loi_oracle.final_order(groups, visit), g["order_new"])


def test_golden_csr_fixture_is_scipy_tocsr_semantics():
    """The CSR the reference's dataset.py built from text == duplicate-merged, sorted scipy CSR
    (pins the input contract the oracle and product assume: ascending unique columns per row)."""
    g = np.load(os.path.join(GOLD, "csr_from_text.npz"))
    for name in ("five_nodes", "rand_40", "rand_333"):
        text = bytes(g[name + "_text"]).decode()
        pairs = [tuple(int(v) for v in ln.split(",")) for ln in text.splitlines()]
        dst = np.array([p[0] for p in pairs]) - 1
        src = np.array([p[1] for p in pairs]) - 1
        n = int(max(dst.max(), src.max())) + 1
        assert n == int(g[name + "_num_nodes"]) and len(pairs) == int(g[name + "_num_edges"])
        m = sp.coo_matrix((np.ones(len(src)), (src, dst)), shape=(n, n)).tocsr()
        m.sum_duplicates()
        m.sort_indices()
        assert np.array_equal(m.indptr, g[name + "_row_pointers"])
        assert np.array_equal(m.indices, g[name + "_column_index"])
        rp, col = g[name + "_row_pointers"], g[name + "_column_index"]
        for r in range(n):
            assert np.all(np.diff(col[rp[r]:rp[r + 1]]) > 0)


def test_config1_example_dataset_against_torch_sparse_mm(oracle_mod):
    """BASELINE config 1 plumbing on CPU: the `example` dataset (hc-spmm_amd/Dataset/example.txt, dim 16)
    through our loader, A*X from the oracle vs torch.sparse.mm (CSR and COO)."""
    import sys
    import torch
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hc-spmm_amd")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    from dataset import HCSPMM_dataset
    ds = HCSPMM_dataset(os.path.join(pkg, "Dataset", "example.txt"), 16, 22, device="cpu", seed=0)
    rp, col = ds.row_pointers.numpy(), ds.column_index.numpy()
    X = ds.x.numpy()
    A = torch.sparse_csr_tensor(ds.row_pointers.long(), ds.column_index.long(), torch.ones(len(col)),
                                size=(ds.num_nodes, ds.num_nodes))
    want = torch.sparse.mm(A, ds.x).numpy()
    got = oracle_mod.spmm_f32(rp, col, X)
    ok, ratio = oracle_mod.check_spmm(want, rp, col, X)
    assert ok, ratio  # torch's own result is within the bar of the fp64 product
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    assert np.abs(torch.sparse.mm(A.to_sparse_coo(), ds.x).numpy() - want).max() <= 1e-5 * np.abs(want).max()
