"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libhcspmm.so, against the CPU oracle on the same seeded inputs.  The matrix runs through BOTH Python
front-ends of that ABI (tests/frontends.py): the ctypes glue `hcspmm` and the torch extension module
`HCSPMM` -- the reference's own boundary (hybrid_all.cpp:500-525), the one GNN_model.py imports.

Bars (north_star): integer products bit-exact; A*X within 1e-5 relative fp32.  Written as:
  * integer-valued X (X[i,:] = i, the reference's gen_test_tensor idea, GNN_model.py:13-23):
    result must be EXACT (every partial sum is an integer < 2^24, so any order gives it);
  * random X: |got - fp64 product| <= 1e-5 * sum_j |x_j| componentwise (oracle.check_spmm), and
    BIT-IDENTICAL to the sequential CSR-order fp32 oracle whenever no row is split (both kernels
    keep CSR order; fp32 MFMA is an exact k-ordered fma chain with 0/1 multipliers).
"""
import numpy as np
import pytest
import torch

import frontends
import hcspmm
from hcspmm import graphs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ctypes", "extension"])
def fe(request):
    return frontends.get(request.param)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: no HIP device visible")
    return torch.device("cuda:0")


def _t(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


class Graph:
    def __init__(self, rp, col, dev, rule=0, plan=True, force_type=None, fe=None, no_slices=False):
        """no_slices: the test asserts CSR-order bits (or plan counts) for EVERY row: build the plan without XCD-affine column
        slices whatever HCSPMM_SLICE_THRESHOLD says (the stress run of this file forces them on everywhere else)."""
        self.fe = fe if fe is not None else frontends.get("ctypes")
        self.rp, self.col = rp, col
        self.N, self.E = len(rp) - 1, len(col)
        self.rp_d, self.col_d = _t(rp, dev), _t(col, dev)
        outs = self.fe.preprocess(self.col_d, self.rp_d, self.N, self.E, (self.N + 15) // 16, rule=rule)
        self.bp, self.e2c, self.e2r, self.ht, self.row_nzr, self.col_nzr = outs
        if force_type is not None:  # force every window onto one sub-path
            self.ht = torch.full_like(self.ht, force_type)
            if plan:  # ... and rebuild the plan for that classification
                self.row_nzr = self.fe.build_plan(self.rp_d, self.col_d, self.bp, self.e2c, self.ht)
        if plan and no_slices:
            self.row_nzr = self.fe.build_plan(self.rp_d, self.col_d, self.bp, self.e2c, self.ht, slice_threshold=-1)
        if not plan:  # the reference's [0] placeholders -> plan-free kernel
            self.row_nzr = torch.zeros(1, dtype=torch.int32, device=dev)

    def args(self):
        return (self.rp_d, self.col_d, self.bp, self.e2c, self.e2r, self.ht, self.row_nzr, self.col_nzr)

    def header(self):
        return self.fe.header(self.row_nzr)

    def forward(self, X, fn=None):
        return (fn or self.fe.forward)(X, *self.args())[0]


def _seq_limit(h, wide_thr):
    """Rows of at most this many entries are summed strictly in CSR order by one lane group: not split into segments,
    not handed to a whole wave, not cut into XCD-affine column slices."""
    lim = min(h.split_threshold, wide_thr)
    return min(lim, h.slice_threshold) if h.n_slices else lim


def _check(oracle_mod, g, X_np, Z, exact_bits=None):
    """Tolerance for every element; bit-identity with the CSR-order fp32 oracle for every row that one
    lane group (or one dense-tile MFMA chain) sums sequentially: all rows on the plan-free kernel,
    rows up to min(split_threshold, wide_threshold) entries on the planned one, dense windows always.
    exact_bits=True additionally demands that NO row falls outside that set."""
    Z = Z.cpu().numpy()
    D = X_np.shape[1]
    ok, ratio = oracle_mod.check_spmm(Z, g.rp, g.col, X_np)
    assert ok, "relative error %.3g x the 1e-5 bar" % ratio
    ref = oracle_mod.spmm_f32(g.rp, g.col, X_np)
    h = g.header()
    deg = np.diff(g.rp)
    if h is None:  # plan-free kernel: fixed whole-wave threshold; dense windows are MFMA chains
        seq = deg <= g.fe.wide_threshold(g.row_nzr, D)
        seq |= np.repeat(g.ht.cpu().numpy() != 0, 16)[:g.N]
    else:
        seq = deg <= _seq_limit(h, g.fe.wide_threshold(g.row_nzr, D))
        seq |= np.repeat(g.ht.cpu().numpy() != 0, 16)[:g.N]
    assert np.array_equal(Z[seq], ref[seq]), "sequentially-summed rows differ from the CSR-order fp32 oracle"
    if exact_bits:
        assert seq.all()


def test_mfma_operand_layout_single_tile(oracle_mod, dev, fe):
    """Pins the v_mfma_f32_16x16x4_f32 operand/accumulator maps on the box: one dense window with an
    ASYMMETRIC 0/1 tile and X rows holding distinct integers per (row, column)."""
    N = 32
    rows = {0: [1, 17, 20], 3: [17], 5: [1, 2, 3, 20, 31], 15: [0], 9: [2, 31]}
    deg = np.zeros(N, np.int64)
    for r, c in rows.items():
        deg[r] = len(c)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.array(rows[r], np.int32) for r in sorted(rows)])
    for D in (16, 64, 20):
        X = (np.arange(N)[:, None] * 100 + np.arange(D)[None, :]).astype(np.float32)
        for plan in (True, False):
            g = Graph(rp, col, dev, plan=plan, force_type=1, fe=fe)
            if plan:
                assert int(g.ht[0]) == 1  # tiny window -> dense-tile path under the intended rule
            Z = g.forward(_t(X, dev)).cpu().numpy()
            assert np.array_equal(Z, oracle_mod.spmm_f32(rp, col, X)), (D, plan)


CASES = [
    # name, generator, split-free?
    ("cora_scale", lambda: graphs.powerlaw_graph(10000, 50000, seed=1), True),     # BASELINE config 2
    ("ragged_hubs", lambda: graphs.powerlaw_graph(1003, 20000, seed=2, max_degree_frac=0.9), False),
    ("uniform", lambda: graphs.uniform_graph(2048, 30000, seed=3), True),
    ("planted_dense", lambda: graphs.planted_dense_graph(1500, seed=4), True),
    ("one_node", lambda: (np.array([0, 1], np.int32), np.array([0], np.int32)), True),
    ("no_edges", lambda: (np.zeros(50, np.int32), np.zeros(0, np.int32)), True),
    ("no_nodes", lambda: (np.zeros(1, np.int32), np.zeros(0, np.int32)), True),  # N = 0: both front-ends return an empty Z
]


@pytest.mark.parametrize("name,gen,split_free", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("D", [32, 128, 256, 16, 22, 7, 96, 1, 300, 1000, 2, 3, 5, 6, 33, 70])
def test_forward_parity_planned(oracle_mod, dev, fe, name, gen, split_free, D):
    rp, col = gen()
    g = Graph(rp, col, dev, fe=fe)
    rng = np.random.default_rng(D)
    X = rng.standard_normal((g.N, D)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)))
    Xi = np.tile((np.arange(g.N, dtype=np.float32) % 4093)[:, None], (1, D))
    Zi = g.forward(_t(Xi, dev)).cpu().numpy()
    assert np.array_equal(Zi, oracle_mod.spmm_f32(rp, col, Xi))


@pytest.mark.parametrize("name,gen,split_free", CASES[:4], ids=[c[0] for c in CASES[:4]])
@pytest.mark.parametrize("D,pad,off", [(22, 0, 0), (22, 3, 1), (32, 5, 3), (7, 2, 1), (64, 1, 1), (130, 7, 2), (5, 0, 0), (4, 1, 1)])
def test_forward_parity_on_views_off_the_16_byte_grid(oracle_mod, dev, name, gen, split_free, D, pad, off):
    """fp32 rows are gathered 16 bytes per lane whatever the width, row stride and base address (element-aligned vectors; a lane
    that would run past the row is moved back onto its last four columns): X and Z as column slices of wider matrices that
    start `off` floats into a row of D + pad floats, both kernels (planned and plan-free), same bits as the contiguous call."""
    import hcspmm
    rp, col = gen()
    g = Graph(rp, col, dev)
    X = np.random.default_rng(D + off).standard_normal((g.N, D)).astype(np.float32)
    Xd = _t(X, dev)
    want = g.forward(Xd)
    _check(oracle_mod, g, X, want)
    Xw = torch.full((g.N, D + pad + off), float("nan"), device=dev)
    Zw = torch.full((g.N, D + pad + off), -7.0, device=dev)
    Xv, Zv = Xw[:, off:off + D], Zw[:, off:off + D]
    Xv.copy_(Xd)
    for row_nzr in (g.row_nzr, torch.zeros(1, dtype=torch.int32, device=dev)):
        Zv.fill_(-7.0)
        hcspmm.forward_into(Xv, Zv, g.rp_d, g.col_d, g.bp, g.e2c, g.e2r, g.ht, row_nzr, g.col_nzr)
        torch.cuda.synchronize()
        if row_nzr is g.row_nzr:
            assert torch.equal(Zv, want)
        else:
            _check(oracle_mod, g, X, Zv.contiguous())
        assert bool((Zw[:, :off] == -7.0).all()) and bool((Zw[:, off + D:] == -7.0).all())  # nothing written outside the slice


@pytest.mark.parametrize("name,gen,split_free", CASES[:4], ids=[c[0] for c in CASES[:4]])
@pytest.mark.parametrize("D", [32, 128, 22])
@pytest.mark.parametrize("mode", ["placeholder", "all_sparse", "all_dense"])
def test_forward_parity_plan_free(oracle_mod, dev, fe, name, gen, split_free, D, mode):
    """The reference calling convention with row_nzr = col_nzr = [0] (plan-free kernel), with the
    classifier's types and with every window forced onto each sub-path."""
    rp, col = gen()
    force = {"placeholder": None, "all_sparse": 0, "all_dense": 1}[mode]
    g = Graph(rp, col, dev, plan=False, force_type=force, fe=fe)
    X = np.random.default_rng(1).standard_normal((g.N, D)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)))  # rows up to 64 entries / dense windows: bit-identical


@pytest.mark.parametrize("name,gen,split_free", CASES[:4], ids=[c[0] for c in CASES[:4]])
@pytest.mark.parametrize("D", [32, 128, 20])
@pytest.mark.parametrize("force", [0, 1])
def test_forward_parity_planned_forced_types(oracle_mod, dev, fe, name, gen, split_free, D, force):
    """Planned kernel with every window forced onto one sub-path.  All-dense exercises dense windows
    far beyond the classifier's reach: hundreds to thousands of condensed columns (the 64-column
    chunk loop of the dense-tile unit) -- still bit-identical to the CSR-order oracle."""
    rp, col = gen()
    g = Graph(rp, col, dev, force_type=force, fe=fe)
    h = g.header()
    assert (h.n_dense > 0 and h.n_tasks == 0) if force else (h.n_dense == 0)
    if force and name != "planted_dense":
        assert h.max_dense_k > 64
    X = np.random.default_rng(2).standard_normal((g.N, D)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)), exact_bits=bool(force))


@pytest.mark.parametrize("rule", [0, 1, 2, 3, 4])
def test_rules_and_aliases(oracle_mod, dev, fe, rule):
    rp, col = graphs.planted_dense_graph(800, seed=9)
    g = Graph(rp, col, dev, rule=rule, fe=fe)
    want = oracle_mod.preprocess(rp, col, rule)
    for w, t in zip(want, (g.bp, g.e2c, g.e2r, g.ht)):
        assert np.array_equal(w, t.cpu().numpy())
    X = np.random.default_rng(2).standard_normal((g.N, 32)).astype(np.float32)
    Xd = _t(X, dev)
    Z0 = g.forward(Xd)
    _check(oracle_mod, g, X, Z0)
    for name in frontends.FORWARD_NAMES:  # every A*X name of the reference's module (hybrid_all.cpp:504-523)
        assert torch.equal(g.forward(Xd, getattr(fe, name)), Z0), name


def test_split_rows_are_deterministic_and_within_tolerance(oracle_mod, dev, fe):
    # star-heavy graph: a few rows with thousands of entries -> segments + fix-up pass
    rp, col = graphs.powerlaw_graph(4000, 120000, seed=5, exponent=1.8, max_degree_frac=0.8)
    assert np.diff(rp).max() > 1024
    g = Graph(rp, col, dev, fe=fe)
    h = g.header()
    assert h.n_split_rows > 0 and h.n_partials >= 2 * h.n_split_rows
    X = np.random.default_rng(3).standard_normal((g.N, 128)).astype(np.float32)
    Xd = _t(X, dev)
    Z1 = g.forward(Xd)
    Z2 = g.forward(Xd)
    assert torch.equal(Z1, Z2)
    _check(oracle_mod, g, X, Z1)  # includes: short rows still bit-identical to the sequential oracle


@pytest.mark.parametrize("name,gen,split_free", CASES[:4], ids=[c[0] for c in CASES[:4]])
@pytest.mark.parametrize("D", [32, 128, 22, 300])
@pytest.mark.parametrize("slices", [(8, 8, 0), (16, 3, 4), (8, 1, 0), (24, 40, 16)], ids=lambda v: "S%d_thr%d_seg%d" % v)
def test_column_slices_parity(oracle_mod, dev, fe, name, gen, split_free, D, slices):
    """XCD-affine column slices forced on small graphs (hcspmm_plan_params.slice_threshold / n_slices; automatic only
    from 65536 columns): rows above the threshold are summed piecewise (1e-5 bar, exact on integer X, run-to-run
    deterministic), rows at or below it keep the CSR-order bits; the result does not depend on which XCD serves a slice."""
    S, thr, seg = slices
    rp, col = gen()
    g = Graph(rp, col, dev, fe=fe)
    g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, slice_threshold=thr, n_slices=S, segment_len=seg,
                              split_threshold=2 * seg)
    h = g.header()
    deg = np.diff(rp)
    sparse_rows = np.repeat(g.ht.cpu().numpy() == 0, 16)[:g.N]
    n_long = int(((deg > thr) & sparse_rows).sum())
    assert h.n_slices == (S if n_long else 0) and h.n_sliced_rows == n_long  # (no row above the threshold: nothing to slice)
    X = np.random.default_rng(D + S).standard_normal((g.N, D)).astype(np.float32)
    Xd = _t(X, dev)
    Z = g.forward(Xd)
    _check(oracle_mod, g, X, Z)
    assert torch.equal(Z, g.forward(Xd))
    Xi = np.tile((np.arange(g.N, dtype=np.float32) % 4093)[:, None], (1, D))
    assert np.array_equal(g.forward(_t(Xi, dev)).cpu().numpy(), oracle_mod.spmm_f32(rp, col, Xi))
    if D == 128:  # 16-bit features through the same sliced plan
        X16 = Xd.to(torch.bfloat16)
        _check_h16(oracle_mod, g, X16, g.forward(X16))


@pytest.mark.parametrize("D", [128, 256, 96])
@pytest.mark.parametrize("panel_cols", [16, 32, 64, 48, -1, 4096])
def test_plan_panel_width_parameter(oracle_mod, dev, fe, D, panel_cols):
    """hcspmm_plan_params.panel_cols: the sparse-row path runs in passes of that many feature columns (any multiple of 16; one
    pass when it covers the embedding); the result keeps the CSR-order bits whatever the width."""
    rp, col = graphs.powerlaw_graph(3000, 60000, seed=12, max_degree_frac=0.2)
    g = Graph(rp, col, dev, fe=fe)
    g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, panel_cols=panel_cols, slice_threshold=-1)
    assert g.header().panel_cols == (-1 if panel_cols < 0 else panel_cols)
    X = np.random.default_rng(D).standard_normal((g.N, D)).astype(np.float32)
    Z = g.forward(_t(X, dev))
    _check(oracle_mod, g, X, Z)
    X16 = _t(X, dev).to(torch.bfloat16)
    _check_h16(oracle_mod, g, X16, g.forward(X16))


def test_tune_plan_returns_the_fastest_measured_variant(oracle_mod, dev):
    rp, col = graphs.powerlaw_graph(20000, 400000, seed=13)
    g = Graph(rp, col, dev)
    plan, report = hcspmm.tune_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.e2r, g.ht, 128, steps=5)
    assert len(report) >= 4 and report == sorted(report, key=lambda r: r["ms"]) and all(r["ms"] > 0 for r in report)
    h = hcspmm.plan_header(plan)
    assert h.panel_cols == report[0]["panel_cols"] and h.num_nodes == g.N
    g.row_nzr = plan
    X = np.random.default_rng(1).standard_normal((g.N, 128)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)))
    # a caller-chosen candidate list
    plan2, rep2 = hcspmm.tune_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.e2r, g.ht, 32, candidates=[dict(slice_threshold=8, n_slices=8), dict()], steps=3)
    assert len(rep2) == 2 and hcspmm.plan_header(plan2).n_slices in (0, 8)


@pytest.mark.parametrize("D", [128, 64, 32, 17, 4])
def test_dense_windows_around_the_compact_record_limit(oracle_mod, dev, fe, D):
    """Dense windows with exactly K = 1 ... 40 (compact 64-word records), 41 ... 80 (128-word records) and 81, 96,
    130 (regular packs) unique columns in one graph, last window ragged (N % 16 != 0)."""
    rng = np.random.default_rng(5)
    Ks = [1, 2, 8, 9, 16, 25, 31, 32, 33, 40, 41, 48, 55, 56, 57, 63, 64, 65, 72, 73, 80, 81, 96, 130, 7]
    N = 16 * len(Ks) - 5
    rows, cols = [], []
    for w, K in enumerate(Ks):
        cset = np.sort(rng.choice(N, K, replace=False))
        nrows = min(16, N - 16 * w)
        m = rng.random((nrows, K)) < 0.4
        m[rng.integers(0, nrows, K), np.arange(K)] = True  # every column used: exactly K unique columns
        r, k = np.nonzero(m)
        rows.append(16 * w + r)
        cols.append(cset[k])
    rp, col = graphs._to_csr(np.concatenate(rows), np.concatenate(cols), N)
    g = Graph(rp, col, dev, force_type=1, fe=fe)
    h = g.header()
    uniq = [len(np.unique(col[rp[16 * w]:rp[min(16 * w + 16, N)]])) for w in range(len(Ks))]
    assert h.n_dense == len(Ks) and h.n_dense_compact == sum(8 * ((u + 7) // 8) <= 40 for u in uniq)
    assert h.n_dense_compact2 == sum(40 < 8 * ((u + 7) // 8) <= 80 for u in uniq)
    X = rng.standard_normal((N, D)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)), exact_bits=True)


@pytest.mark.parametrize("D", [128, 32])
def test_composite_graph_threaded_plan(oracle_mod, dev, fe, D):
    """70 K rows: compact dense windows, tiny / ordinary / wide / split sparse rows in one launch, plan built by
    the multi-threaded host path (>= 4096 windows)."""
    rp, col = graphs.planted_dense_graph_fast(70000, seed=8, dense_fraction=0.4, k_cols=12, fill=0.5, sparse_degree=3)
    N = len(rp) - 1
    rng = np.random.default_rng(8)
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
    extra_r = np.repeat(np.array([17, 30011, 69990]), [700, 1300, 520])
    rp, col = graphs._to_csr(np.concatenate([rows, extra_r]),
                             np.concatenate([col.astype(np.int64), rng.integers(0, N, extra_r.shape[0])]), N)
    g = Graph(rp, col, dev, fe=fe)
    h = g.header()
    assert h.n_dense_compact > 0 and h.n_tiny > 0 and h.n_split_rows >= 2 and h.n_tasks > h.n_tiny
    X = rng.standard_normal((N, D)).astype(np.float32)
    _check(oracle_mod, g, X, g.forward(_t(X, dev)))
    X16 = torch.from_numpy(X).to(torch.bfloat16).to(dev)
    _check_h16(oracle_mod, g, X16, g.forward(X16))


_H16_GRAPHS = [
    ("powerlaw", lambda: graphs.powerlaw_graph(2000, 40000, seed=31, max_degree_frac=0.3)),   # wide + split rows
    ("planted", lambda: graphs.planted_dense_graph(1500, seed=32)),                            # compact dense windows
    ("uniform", lambda: graphs.uniform_graph(1003, 9000, seed=33)),                            # N % 16 != 0
    ("tiny", lambda: _tiny_graph()),
]


def _check_h16(oracle_mod, g, X16, Z16):
    """16-bit features: Z = round_dtype(fp32 sum in the fp32 path's order).  Bit-identical to the rounded CSR-order
    fp32 oracle on every row that is summed sequentially; elsewhere within 1e-5 * sum|x| of the fp64 product plus one
    rounding of the result."""
    dtype = X16.dtype
    assert Z16.dtype == dtype
    Xf = X16.float().cpu().numpy()
    D = Xf.shape[1]
    ref32 = oracle_mod.spmm_f32(g.rp, g.col, Xf)
    want = torch.from_numpy(ref32).to(dtype)  # torch rounds to nearest even, like the kernel
    h = g.header()
    deg = np.diff(g.rp)
    thr = g.fe.wide_threshold(g.row_nzr, D, dtype)
    seq = deg <= (thr if h is None else _seq_limit(h, thr))
    seq |= np.repeat(g.ht.cpu().numpy() != 0, 16)[:g.N]
    got = Z16.cpu()
    assert torch.equal(got[torch.from_numpy(seq)].view(torch.int16), want[torch.from_numpy(seq)].view(torch.int16))
    ref64 = oracle_mod.spmm_f64(g.rp, g.col, Xf)
    mag = oracle_mod.spmm_f64(g.rp, g.col, Xf, absolute=True)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    assert np.all(np.abs(got.double().numpy() - ref64) <= 1e-5 * mag + eps * np.abs(ref64) + 1e-30)
    return seq


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("D", [256, 128, 64, 40, 32, 20, 7, 22, 6, 10, 70, 130, 4])
@pytest.mark.parametrize("name,gen", _H16_GRAPHS, ids=[g[0] for g in _H16_GRAPHS])
def test_half_precision_features_planned(oracle_mod, dev, fe, name, gen, D, dtype):
    rp, col = gen()
    g = Graph(rp, col, dev, fe=fe)
    X16 = torch.from_numpy(np.random.default_rng(D).standard_normal((g.N, D)).astype(np.float32)).to(dtype).to(dev)
    seq = _check_h16(oracle_mod, g, X16, g.forward(X16))
    assert seq.any()
    # integer-valued features small enough for the format: exact whatever the order
    Xi = ((torch.arange(g.N)[:, None] + torch.arange(D)[None, :]) % 4).to(dtype).to(dev)
    Zi = g.forward(Xi).float().cpu().numpy()
    assert np.array_equal(Zi, torch.from_numpy(oracle_mod.spmm_f32(rp, col, Xi.float().cpu().numpy())).to(dtype).float().numpy())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("D", [128, 24, 5, 22, 70, 134])
@pytest.mark.parametrize("mode", ["plan_free", "all_dense", "all_sparse"])
def test_half_precision_features_other_paths(oracle_mod, dev, fe, D, dtype, mode):
    rp, col = graphs.powerlaw_graph(1500, 20000, seed=41, max_degree_frac=0.2)
    g = Graph(rp, col, dev, plan=(mode != "plan_free"), force_type={"all_dense": 1, "all_sparse": 0}.get(mode), fe=fe)
    X16 = torch.from_numpy(np.random.default_rng(7).standard_normal((g.N, D)).astype(np.float32)).to(dtype).to(dev)
    _check_h16(oracle_mod, g, X16, g.forward(X16))


def test_half_precision_strided_views_and_errors(oracle_mod, dev, fe):
    rp, col = graphs.powerlaw_graph(900, 12000, seed=43)
    g = Graph(rp, col, dev, fe=fe)
    wide = torch.randn(g.N, 192, device=dev).to(torch.bfloat16)
    out = torch.zeros(g.N, 192, dtype=torch.bfloat16, device=dev)
    fe.forward_into(wide[:, 64:128], out[:, 128:192], *g.args())
    assert torch.equal(out[:, 128:192], g.forward(wide[:, 64:128].contiguous())) and not out[:, :128].any()
    # views whose first element is only 8- / 4- / 2-byte aligned: 16-byte lanes on the dword grid, and the 1-element-per-lane build
    for off in (4, 2, 6, 1):
        out.zero_()
        fe.forward_into(wide[:, off:off + 64], out[:, off:off + 64], *g.args())
        assert torch.equal(out[:, off:off + 64], g.forward(wide[:, off:off + 64].contiguous()))
        assert not out[:, :off].any() and not out[:, off + 64:].any()
    # a width that is not a multiple of 8, in a view off the 16-byte grid: three overlapping 16-byte lanes per row
    for off, D in ((2, 22), (6, 70), (0, 10)):
        out.zero_()
        fe.forward_into(wide[:, off:off + D], out[:, off:off + D], *g.args())
        assert torch.equal(out[:, off:off + D], g.forward(wide[:, off:off + D].contiguous()))
        assert not out[:, :off].any() and not out[:, off + D:].any()
    with pytest.raises(RuntimeError, match="float32 / float16 / bfloat16"):
        fe.forward_into(wide[:, :64], torch.zeros(g.N, 64, device=dev), *g.args())  # mixed dtypes
    with pytest.raises(RuntimeError, match="float32"):
        g.forward(torch.zeros(g.N, 8, dtype=torch.float64, device=dev))
    with pytest.raises(RuntimeError, match="float32"):  # the fused update stays fp32
        fe.forward_fixed32_fused(wide[:, :32].contiguous(), *g.args(), torch.zeros(32, 8, device=dev))


def _tiny_graph(N=3000, seed=9):
    """Mostly rows of 0, 1 and 2 entries (the descriptors that carry their indices inline), a sprinkling of
    longer ones, and hub rows of 513 / 514 / 770 entries whose LAST segment has 1 / 2 / 2 entries."""
    rng = np.random.default_rng(seed)
    deg = rng.choice([0, 1, 2, 3, 7, 40], size=N, p=[0.35, 0.3, 0.2, 0.1, 0.04, 0.01])
    deg[[5, 1700, 2999]] = [513, 514, 770]
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg if d]).astype(np.int32)
    return rp, col


@pytest.mark.parametrize("D", [128, 64, 32, 20, 16, 6, 3, 1])
def test_tiny_tasks_and_tiny_segments(oracle_mod, dev, fe, D):
    rp, col = _tiny_graph()
    g = Graph(rp, col, dev, rule=2, fe=fe, no_slices=True)  # every window on the sparse-row path
    h = g.header()
    deg = np.diff(rp)
    assert h.n_tiny == int((deg <= 2).sum()) + 3 and h.n_split_rows == 3
    X = np.random.default_rng(D).standard_normal((g.N, D)).astype(np.float32)
    Z = g.forward(_t(X, dev))
    _check(oracle_mod, g, X, Z)
    Xi = (np.arange(g.N, dtype=np.float32)[:, None] + np.arange(D, dtype=np.float32)[None, :] % 3)
    assert np.array_equal(g.forward(_t(Xi, dev)).cpu().numpy(), oracle_mod.spmm_f32(rp, col, Xi))


_FUSED_GRAPHS = [
    ("powerlaw", lambda: graphs.powerlaw_graph(1200, 9000, seed=6)),
    ("planted_ragged", lambda: graphs.planted_dense_graph(1500 - 7, seed=14)),  # compact dense windows, N % 16 != 0
    ("wide_windows", lambda: _wide_window_graph()),                             # K = 48..130: double records + regular packs
    ("hubs", lambda: graphs.powerlaw_graph(3000, 30000, seed=9, max_degree_frac=0.3)),  # hub rows: wide / split / sliced
]


def _wide_window_graph(seed=11):
    rng = np.random.default_rng(seed)
    Ks = [48, 56, 64, 80, 88, 96, 130, 24, 8, 130, 72, 41]
    N = 16 * len(Ks)
    rows, cols = [], []
    for w, K in enumerate(Ks):
        cset = np.sort(rng.choice(N, K, replace=False))
        m = rng.random((16, K)) < 0.5
        m[rng.integers(0, 16, K), np.arange(K)] = True
        r, k = np.nonzero(m)
        rows.append(16 * w + r)
        cols.append(cset[k])
    return graphs._to_csr(np.concatenate(rows), np.concatenate(cols), N)


@pytest.mark.parametrize("D,H", [(32, 32), (64, 64), (96, 22), (32, 7), (16, 40), (128, 32), (48, 16), (256, 32), (16, 16),
                                 (96, 16), (128, 64), (64, 32), (32, 16), (512, 32), (22, 32), (22, 22), (32, 96), (6, 50),
                                 (32, 22), (64, 22), (64, 50), (128, 22), (32, 2), (48, 31), (32, 40), (50, 32), (70, 16), (18, 7), (100, 22)])
@pytest.mark.parametrize("gname,gen", _FUSED_GRAPHS, ids=[g[0] for g in _FUSED_GRAPHS])
@pytest.mark.parametrize("form", ["two_launches", "in_launch", "row_tiles"])
def test_fused_variants(oracle_mod, dev, fe, gname, gen, D, H, form):
    """out2 = A*X, out = out2 * W for every fused name of the reference's module, in the three forms of the operator: the
    default (hybrid launch + update launch); for plans built with fuse_in_launch = 1, dense-tile windows updated inside
    the hybrid launch (transposed-tile MFMA chain) where the shape allows; for fuse_in_launch = 2 the sparse-row path as
    well (tiles of 16 tasks summed, parked in LDS and multiplied in one launch -- `out` must have the two-launch form's
    BITS).  The wide_windows graph is forced all-dense so that every record kind takes the in-launch path; the hubs graph
    is forced all-sparse with small split / slice thresholds so that the rows the row-tile launch leaves out (whole-wave
    rows, split rows, sliced rows in one piece) go through the leftover update."""
    rp, col = gen()
    g = Graph(rp, col, dev, fe=fe, force_type={"wide_windows": 1, "hubs": 0}.get(gname))
    assert not hcspmm.fused_in_launch(g.row_nzr, 32, 32)  # the default plan: two launches
    kw = dict(split_threshold=48, segment_len=16, slice_threshold=20) if gname == "hubs" else {}
    if kw:
        g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, **kw)
        assert g.header().n_split_rows > 0 and g.header().n_slices > 0
    Xr = torch.randn(g.N, D, device=dev)
    Wr = torch.randn(D, H, device=dev)
    out_two = fe.forward_fixed32_fused(Xr, *g.args(), Wr)[0]  # the two-launch form on the same task cut
    asked = {"two_launches": 0, "in_launch": 1, "row_tiles": 2}[form]
    if asked:  # (row tiles need the sparse region in one column pass: automatic below 64 columns and on short-row graphs)
        g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=asked, panel_cols=-1 if asked == 2 else 0, **kw)
    h = g.header()
    # in the launch: asked for, fp32, D a multiple of 16 from 32 up, H = 16 or 32 (output tiles held in registers), W fits the LDS staging area
    dense_ok = D % 16 == 0 and D >= 32 and H in (16, 32) and D * (H + 4) * 4 <= 64 * 1024
    chunked = D == 128  # two column chunks of 64 summed one after the other, eight waves per workgroup
    Hp = (H + 15) // 16 * 16  # hidden widths between the tile sizes are zero-padded to one, two or four 16-column output tiles
    Dp = (D + 15) // 16 * 16  # ... and input widths to the next multiple of 16 (zero rows in the staged weights, zero tile columns)
    lds = (Dp * (Hp + 4) + (8 if chunked else 4) * (16 * ((64 if chunked else Dp) + 4) + 16)) * 4
    rows_ok = 17 <= D <= 128 and 1 <= H <= 64 and Hp != 48 and lds <= 64 * 1024
    assert hcspmm.fused_in_launch(g.row_nzr, D, H) == (2 if asked == 2 and rows_ok else (1 if asked == 1 and h.n_dense > 0 and dense_ok else 0))
    if gname not in ("powerlaw", "hubs"):
        assert h.n_dense > 0
    if asked == 2:  # the update is the streaming kernel's MFMA chain, fed from LDS instead of HBM: the two-launch form's bits
        assert torch.equal(fe.forward_fixed32_fused(Xr, *g.args(), Wr)[0], out_two)
    rng = np.random.default_rng(7)
    X = rng.standard_normal((g.N, D)).astype(np.float32)
    W = rng.standard_normal((D, H)).astype(np.float32)
    want_out, want_out2 = oracle_mod.spmm_fused_f32(rp, col, X, W)
    Xd, Wd = _t(X, dev), _t(W, dev)
    Z_plain = g.forward(Xd)
    scale = oracle_mod.spmm_f64(rp, col, X, absolute=True) @ np.abs(W).astype(np.float64)  # sum|x_j| . |W|
    for name in frontends.FUSED_NAMES:
        out, out2 = getattr(fe, name)(Xd, *g.args(), Wd)
        assert torch.equal(out2, Z_plain), name  # the aggregate has the unfused operator's bits, whichever form ran
        assert oracle_mod.check_spmm(out2.cpu().numpy(), rp, col, X)[0]
        assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30), name
        assert torch.equal(out, getattr(fe, name)(Xd, *g.args(), Wd)[0])  # deterministic
    # transposed (non-contiguous) weights, as the reference's backward passes them (GNN_model.py:98,120)
    Wt_d = _t(np.ascontiguousarray(W.T), dev).transpose(0, 1)
    assert not Wt_d.is_contiguous()
    out, out2 = fe.forward_fixed32_fused(Xd, *g.args(), Wt_d)
    assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30)
    # final_fused writes the caller's output buffer and returns it
    for name in frontends.FINAL_FUSED_NAMES:
        buf = torch.zeros(g.N, H, device=dev)
        out, out2 = getattr(fe, name)(Xd, *g.args(), Wd, buf)
        assert out.data_ptr() == buf.data_ptr()
        assert np.all(np.abs(buf.cpu().numpy().astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30)
    # integer-valued X and W: every product and partial sum is exact, so out must be EXACT in either form
    Xi = (np.arange(g.N, dtype=np.float32)[:, None] % 7 + np.arange(D, dtype=np.float32)[None, :] % 3)
    Wi = ((np.arange(D)[:, None] + 2 * np.arange(H)[None, :]) % 5 - 2).astype(np.float32)
    out, out2 = fe.forward_fixed32_fused(_t(Xi, dev), *g.args(), _t(Wi, dev))
    zi = oracle_mod.spmm_f32(rp, col, Xi)
    assert np.array_equal(out2.cpu().numpy(), zi) and np.array_equal(out.cpu().numpy(), zi.astype(np.float64) @ Wi.astype(np.float64))


@pytest.mark.parametrize("D,H", [(32, 22), (22, 32), (32, 32), (64, 64), (32, 16), (22, 22), (32, 7)])
def test_fused_final_into_an_output_that_is_only_4_byte_aligned(oracle_mod, dev, D, H):
    """forward_final_fused writes the caller's `output`: a matrix that starts one float into an allocation (rows of H floats, so
    neither the base nor the rows sit on the 8- / 16-byte grid) takes the same vector stores with element alignment -- same bits
    as an aligned output, nothing written outside it."""
    import hcspmm
    rp, col = graphs.planted_dense_graph(1500 - 7, seed=14)
    g = Graph(rp, col, dev)
    X, W = torch.randn(g.N, D, device=dev), torch.randn(D, H, device=dev)
    want = hcspmm.forward_fixed32_fused(X, *g.args(), W)[0]
    for form in (0, 2):
        args = list(g.args())
        if form:
            args[6] = hcspmm.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=2, panel_cols=-1)
        raw = torch.full((g.N * H + 3,), -3.0, device=dev)
        buf = raw[1:1 + g.N * H].view(g.N, H)
        assert buf.data_ptr() % 8 == 4
        out, out2 = hcspmm.forward_final_fused(X, *args, W, buf)
        torch.cuda.synchronize()
        assert out.data_ptr() == buf.data_ptr() and torch.equal(buf, want)
        assert float(raw[0]) == -3.0 and bool((raw[1 + g.N * H:] == -3.0).all())


@pytest.mark.parametrize("D,H", [(32, 32), (64, 16), (64, 64), (128, 32), (96, 64)])
def test_fused_row_tiles_are_automatic_on_million_row_graphs(oracle_mod, dev, fe, D, H):
    """A graph whose aggregate (N x D fp32) is 80 MB or more takes the row-tile form without being asked (out2 is then beyond
    what the update launch finds in the caches); a plan built with fuse_in_launch = -1 keeps two launches.  Same bits in out2 AND out; exact
    integer checksums over every row of both sub-paths; sampled rows against the oracle.  (Composite graph: compact dense
    windows, tiny / ordinary / wide / split / column-sliced rows.)"""
    rp, col = graphs.planted_dense_graph_fast(1100000, seed=5, dense_fraction=0.4, k_cols=12, fill=0.5, sparse_degree=3)
    N = len(rp) - 1
    rng = np.random.default_rng(5)
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
    hub_r = np.repeat(np.array([17, 300011, 1099990, 555555]), [700, 1300, 520, 12000])
    picks = rng.choice(N, 6000, replace=False)
    long_r = np.repeat(picks[:3000], 300)  # rows beyond the slice threshold, holding a third of the sparse entries: slices on
    mid_r = np.repeat(picks[3000:], 40)    # ordinary tasks of several length classes
    extra_r = np.concatenate([hub_r, long_r, mid_r])
    rp, col = graphs._to_csr(np.concatenate([rows, extra_r]),
                             np.concatenate([col.astype(np.int64), rng.integers(0, N, extra_r.shape[0])]), N)
    g = Graph(rp, col, dev, fe=fe)
    h = g.header()
    assert h.n_dense > 0 and h.n_tiny > 0 and h.n_split_rows > 0 and h.n_slices > 0 and h.n_tasks > h.n_tiny
    if D > 64 and H > 32:  # beyond 64 columns the form is automatic for H <= 32 on dense-heavy graphs (this one: 40 % of the windows), else opt-in
        assert hcspmm.fused_in_launch(g.row_nzr, D, H) == 0
        g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=2)
    assert hcspmm.fused_in_launch(g.row_nzr, D, H) == 2
    never = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=-1)
    assert hcspmm.fused_in_launch(never, D, H) == 0
    X = torch.randn(N, D, device=dev)
    W = torch.randn(D, H, device=dev)
    out, out2 = fe.forward_fixed32_fused(X, *g.args(), W)
    args0 = g.args()[:6] + (never, g.col_nzr)
    out_0, out2_0 = fe.forward_fixed32_fused(X, *args0, W)
    assert torch.equal(out2, out2_0) and torch.equal(out, out_0)
    assert torch.equal(out2, g.forward(X))
    assert torch.equal(out, fe.forward_fixed32_fused(X, *g.args(), W)[0])  # deterministic
    # integer-valued X and W: exact
    ids = (np.arange(N) % 7).astype(np.float32)  # (every partial sum of out stays below 2^24: 12 000 x 8 x 2 D)
    Xi = np.tile(ids[:, None], (1, D)) + (np.arange(D, dtype=np.float32) % 3)[None, :]
    Wi = ((np.arange(D)[:, None] + 2 * np.arange(H)[None, :]) % 5 - 2).astype(np.float32)
    oi, o2i = fe.forward_fixed32_fused(_t(Xi, dev), *g.args(), _t(Wi, dev))
    cs = np.concatenate([[0.0], np.cumsum(ids[col].astype(np.float64))])
    deg = np.diff(rp).astype(np.float64)
    want2 = (cs[rp[1:]] - cs[rp[:-1]])[:, None] + deg[:, None] * (np.arange(D) % 3)[None, :]
    assert np.array_equal(o2i.cpu().numpy(), want2.astype(np.float32))
    assert np.array_equal(oi.cpu().numpy(), (want2 @ Wi.astype(np.float64)).astype(np.float32))
    # sampled rows vs the oracle
    rs = np.concatenate([rng.choice(N, 2000, replace=False), [17, 300011, 555555]])
    sub_rp = np.concatenate([[0], np.cumsum(np.diff(rp)[rs])]).astype(np.int32)
    sub_col = np.concatenate([col[rp[r]:rp[r + 1]] for r in rs]).astype(np.int32)
    Xh, Wh = X.cpu().numpy(), W.cpu().numpy()
    assert oracle_mod.check_spmm(out2.cpu().numpy()[rs], sub_rp, sub_col, Xh)[0]
    want_out, _ = oracle_mod.spmm_fused_f32(sub_rp, sub_col, Xh, Wh)
    scale = oracle_mod.spmm_f64(sub_rp, sub_col, Xh, absolute=True) @ np.abs(Wh).astype(np.float64)
    assert np.all(np.abs(out.cpu().numpy()[rs].astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30)


@pytest.mark.parametrize("N,D,H", [(70001, 96, 32), (5000, 32, 32), (9999, 32, 22), (4097, 64, 64), (300, 7, 3),
                                   (66000, 128, 32), (1, 16, 16), (20000, 80, 48)])
def test_weight_grad_kernel(dev, fe, N, D, H):
    """dW = A^T B (split-K MFMA): against the fp64 product, tolerance 1e-5 * sum_n |a||b| per element (the same kind of
    bar as A*X); deterministic; strided inputs; unsupported shapes decline."""
    rng = np.random.default_rng(N + D)
    A = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).to(dev)
    B = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32)).to(dev)
    got = fe.weight_grad(A, B)
    assert got is not None and got.shape == (D, H)
    ref = A.double().t() @ B.double()
    scale = A.double().abs().t() @ B.double().abs()
    assert bool(((got.double() - ref).abs() <= 1e-5 * scale + 1e-30).all())
    assert torch.equal(got, fe.weight_grad(A, B))
    wideA, wideB = torch.zeros(N, D + 5, device=dev), torch.zeros(N, H + 3, device=dev)
    wideA[:, 2:2 + D], wideB[:, 1:1 + H] = A, B
    assert torch.equal(fe.weight_grad(wideA[:, 2:2 + D], wideB[:, 1:1 + H]), got)


def test_weight_grad_declines_unsupported(dev):
    A, B = torch.zeros(100, 200, device=dev), torch.zeros(100, 16, device=dev)
    assert hcspmm.weight_grad(A, B) is None                      # more than 8 row tiles
    assert hcspmm.weight_grad(A[:, :96], torch.zeros(100, 80, device=dev)) is None  # more than 4 column tiles
    assert hcspmm.weight_grad(A[:, :128], torch.zeros(100, 64, device=dev)) is None  # 8 x 4 tiles: too many accumulators
    assert hcspmm.weight_grad(A[:, :16].double(), B.double()) is None


def test_error_behaviour(oracle_mod, dev, fe):
    rp, col = graphs.powerlaw_graph(200, 900, seed=1)
    g = Graph(rp, col, dev, fe=fe, no_slices=True)
    X = torch.zeros(g.N, 16, device=dev)
    with pytest.raises(RuntimeError, match="input must be contiguous"):
        fe.forward(X.t().contiguous().t(), *g.args())
    with pytest.raises(RuntimeError, match="nodePointer must be a CUDA tensor"):
        fe.forward(X, g.rp_d.cpu(), *g.args()[1:])
    with pytest.raises(RuntimeError, match="rows"):
        fe.forward(torch.zeros(g.N + 1, 16, device=dev), *g.args())
    # a plan built for another graph is refused (HCSPMM_EPLAN), not silently used
    other = Graph(*graphs.powerlaw_graph(300, 900, seed=2), dev, fe=fe)
    with pytest.raises(RuntimeError, match="plan"):
        fe.forward(X, g.rp_d, g.col_d, g.bp, g.e2c, g.e2r, g.ht, other.row_nzr, g.col_nzr)
    # ... also when it has the same N and E (a relabelled graph does): told apart by the graph fingerprint
    perm = torch.randperm(g.N, generator=torch.Generator().manual_seed(3)).to(torch.int32)
    rp2, col2 = hcspmm.apply_permutation(torch.from_numpy(rp), torch.from_numpy(col), perm)
    twin = Graph(rp2.numpy(), col2.numpy(), dev, fe=fe)
    assert (twin.N, twin.E) == (g.N, g.E)
    with pytest.raises(RuntimeError, match="plan does not match this graph"):
        fe.forward(X, g.rp_d, g.col_d, g.bp, g.e2c, g.e2r, g.ht, twin.row_nzr, g.col_nzr)
    # clones of the SAME graph's tensors (new addresses) pass the fingerprint check -- once, then cached
    Xr = torch.randn(g.N, 16, device=dev)
    args = [t.clone() for t in g.args()]
    for _ in range(2):
        Z = fe.forward(Xr, *args)[0]
        assert np.array_equal(Z.cpu().numpy(), oracle_mod.spmm_f32(rp, col, Xr.cpu().numpy()))
    # a column id beyond X: refused on the host at preprocess, never launched
    bad = col.copy()
    bad[len(bad) // 2] = g.N
    with pytest.raises(RuntimeError, match="invalid argument"):
        fe.preprocess(_t(bad, dev), g.rp_d, g.N, g.E, (g.N + 15) // 16)
    # a row block whose columns index a taller X: forward() wants a square product, forward_rect() takes it
    tall = fe.preprocess(g.col_d, g.rp_d, g.N, g.E, (g.N + 15) // 16, num_columns=2 * g.N)
    with pytest.raises(RuntimeError, match="rows"):
        fe.forward(Xr, g.rp_d, g.col_d, *tall)
    tall = list(tall)
    tall[4] = fe.build_plan(g.rp_d, g.col_d, tall[0], tall[1], tall[3], num_columns=2 * g.N, slice_threshold=-1)  # (bits compared with Z)
    Zt = fe.forward_rect(torch.cat([Xr, Xr]), g.rp_d, g.col_d, *tall)[0]
    assert torch.equal(Zt, Z)


def test_reddit_scale_properties(oracle_mod, dev):
    """BASELINE config 3 at full size (233K nodes / 11.6M entries, D = 128): too big for the scalar
    oracle to be quick, so parity is checked through size-independent properties -- exact integer
    checksums (X = 1 -> degrees; X[i,:] = i mod 1021), linearity, a fixed sample of rows against
    the oracle, and run-to-run determinism."""
    rp, col = graphs.powerlaw_graph(233000, 11600000, seed=3)
    g = Graph(rp, col, dev)
    N, D = g.N, 128
    deg = np.diff(rp).astype(np.float32)
    Z = g.forward(torch.ones(N, D, device=dev)).cpu().numpy()
    assert np.array_equal(Z, np.tile(deg[:, None], (1, D)))
    ids = (np.arange(N) % 1021).astype(np.float32)
    Zi = g.forward(_t(np.tile(ids[:, None], (1, D)), dev)).cpu().numpy()
    cs = np.concatenate([[0.0], np.cumsum(ids[col].astype(np.float64))])
    want = cs[rp[1:]] - cs[rp[:-1]]
    assert np.array_equal(Zi[:, 0], want.astype(np.float32)) and np.array_equal(Zi[:, 0], Zi[:, D - 1])
    rng = np.random.default_rng(0)
    X1 = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).to(dev)
    X2 = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32)).to(dev)
    Z1, Z2, Z12 = g.forward(X1), g.forward(X2), g.forward(X1 + 2 * X2)
    lin = (Z12 - (Z1 + 2 * Z2)).abs().max().item() / Z12.abs().max().item()
    assert lin < 1e-5
    assert torch.equal(Z1, g.forward(X1))
    # sampled rows vs the oracle (sub-graph made of the sampled rows only)
    rows = rng.choice(N, 2000, replace=False)
    sub_deg = np.diff(rp)[rows]
    sub_rp = np.concatenate([[0], np.cumsum(sub_deg)]).astype(np.int32)
    sub_col = np.concatenate([col[rp[r]:rp[r + 1]] for r in rows]).astype(np.int32)
    ok, ratio = oracle_mod.check_spmm(Z1.cpu().numpy()[rows], sub_rp, sub_col, X1.cpu().numpy())
    assert ok, ratio


def test_forward_is_hip_graph_capturable(oracle_mod, dev, fe):
    """hcspmm_forward neither synchronises nor allocates, so a caller may capture it (both launches:
    hybrid kernel + fix-up) into a HIP graph and replay it."""
    rp, col = graphs.powerlaw_graph(3000, 60000, seed=8, max_degree_frac=0.5)
    g = Graph(rp, col, dev, fe=fe)
    assert g.header().n_split_rows > 0
    X = np.random.default_rng(4).standard_normal((g.N, 64)).astype(np.float32)
    Xd = _t(X, dev)
    g.forward(Xd)  # warm-up outside capture (registers the plan, loads code objects)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            Z = g.forward(Xd)
    Xd.copy_(_t(2 * X, dev))
    graph.replay()
    torch.cuda.synchronize()
    _check(oracle_mod, g, 2 * X, Z)


def test_fused_row_tiles_are_hip_graph_capturable(oracle_mod, dev, fe):
    """The row-tile form of the fused operators is five launches (tile launches, hybrid remainder, fix-up, leftover update) and
    nothing else -- no synchronisation, no allocation outside torch's allocator, the occupancy query cached by the warm-up
    call -- so a training step that contains it captures into a HIP graph (HC-SpMM_main.py --graph) and replays on new inputs."""
    rp, col = graphs.powerlaw_graph(3000, 60000, seed=8, max_degree_frac=0.5)
    g = Graph(rp, col, dev, fe=fe)
    g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=2, split_threshold=64, segment_len=32, slice_threshold=40)
    h = g.header()
    assert hcspmm.fused_in_launch(g.row_nzr, 32, 32) == 2 and h.n_split_rows > 0 and h.n_slices > 0
    rng = np.random.default_rng(4)
    X, W = rng.standard_normal((g.N, 32)).astype(np.float32), rng.standard_normal((32, 32)).astype(np.float32)
    Xd, Wd = _t(X, dev), _t(W, dev)
    fe.forward_fixed32_fused(Xd, *g.args(), Wd)  # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            out, out2 = fe.forward_fixed32_fused(Xd, *g.args(), Wd)
    Xd.copy_(_t(2 * X, dev))
    graph.replay()
    torch.cuda.synchronize()
    want_out, want_out2 = oracle_mod.spmm_fused_f32(rp, col, 2 * X, W)
    _check(oracle_mod, g, 2 * X, out2)
    scale = oracle_mod.spmm_f64(rp, col, 2 * X, absolute=True) @ np.abs(W).astype(np.float64)
    assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30)


def test_offsets_beyond_2_to_31_elements(dev):
    """X and Z with more than 2^31 elements (BASELINE config 5 territory: 16 M nodes x 128): row offsets
    must be 64-bit.  9 M nodes x 256 columns = 2.3e9 elements (9.2 GB each); checked with exact
    integer checksums on both sub-paths (columns chosen from the top of the id range)."""
    N, D = 9_000_000, 256
    rng = np.random.default_rng(5)
    # sparse rows: 4 entries each, biased to high column ids; plus planted dense windows at the end
    deg = np.full(N, 4, np.int64)
    cols = (N - 1 - rng.integers(0, N // 3, size=4 * N)).astype(np.int64).reshape(N, 4)
    cols.sort(axis=1)
    for k in range(1, 4):  # make entries unique within a row
        cols[:, k] = np.maximum(cols[:, k], cols[:, k - 1] + 1)
    cols = np.minimum(cols, N - 1)
    first_planted = (N // 16 - 100_000) * 16  # the last 100 K windows: all 16 rows share the window's first row's columns
    cols[first_planted:] = np.repeat(cols[first_planted::16], 16, axis=0)
    keep = np.ones((N, 4), bool)
    keep[:, 1:] = cols[:, 1:] > cols[:, :-1]
    rp = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
    col = cols[keep].astype(np.int32)
    g = Graph(rp, col, dev)
    h = g.header()
    assert h.n_dense >= 99_000 and h.n_tasks > 7_000_000  # planted windows go dense, the rest sparse
    ids = torch.arange(N, device=dev, dtype=torch.float32) % 1021
    X = ids[:, None].expand(N, D).contiguous()
    assert X.numel() > 2 ** 31
    Z = g.forward(X)
    want = np.zeros(N, np.float64)
    np.add.at(want, np.repeat(np.arange(N), np.diff(rp)), (col.astype(np.int64) % 1021).astype(np.float64))
    want_t = torch.from_numpy(want.astype(np.float32)).to(dev)
    assert torch.equal(Z[:, 0], want_t) and torch.equal(Z[:, D - 1], want_t) and torch.equal(Z[:, 100], want_t)
    del X, Z
    torch.cuda.empty_cache()


def test_forward_into_strided_views(oracle_mod, dev, fe):
    """Strided operator: read a column panel of a wider X, write a column panel of a wider Z, in place."""
    rp, col = graphs.planted_dense_graph(1500, seed=4)  # both sub-paths
    g = Graph(rp, col, dev, fe=fe, no_slices=True)
    rng = np.random.default_rng(9)
    Xw = rng.standard_normal((g.N, 160)).astype(np.float32)
    Xd = _t(Xw, dev)
    Zd = torch.full((g.N, 200), -7.0, device=dev)
    for (x0, z0, w) in ((32, 64, 64), (0, 0, 32), (96, 136, 20)):
        fe.forward_into(Xd[:, x0:x0 + w], Zd[:, z0:z0 + w], *g.args())
        ref = oracle_mod.spmm_f32(rp, col, np.ascontiguousarray(Xw[:, x0:x0 + w]))
        got = Zd.cpu().numpy()
        assert np.array_equal(got[:, z0:z0 + w], ref)
        untouched = np.ones(200, bool)
        untouched[z0:z0 + w] = False
        Zd[:, z0:z0 + w] = -7.0
        assert np.all(got[:, untouched] == -7.0)
    with pytest.raises(RuntimeError, match="unit inner stride"):
        fe.forward_into(Xd.t()[:160, :g.N].t()[:, ::2], Zd[:, :80], *g.args())


def _full_size_properties(oracle_mod, dev, rp, col, n_cols, D, fe=None):
    """Parity at sizes the scalar oracle cannot finish quickly, through size-independent properties (the pattern of
    test_reddit_scale_properties): exact integer checksums over EVERY row (X = 1 -> degrees; X[i,:] = i mod m with m
    small enough that every partial sum stays below 2^24, so any summation order is exact), run-to-run determinism,
    and 2 000 sampled rows against the oracle.  A is n x n_cols (n_cols > n: one GPU's row block of a sharded graph,
    every X row resident).  Returns the plan header."""
    fe = fe or frontends.get("ctypes")
    N, E = len(rp) - 1, len(col)
    rp_d, col_d = _t(rp, dev), _t(col, dev)
    outs = fe.preprocess(col_d, rp_d, N, E, (N + 15) // 16, rule=0, num_columns=n_cols)
    h = fe.header(outs[4])
    assert h.num_columns == n_cols and h.nnz_sparse + h.nnz_dense == E
    deg = np.diff(rp)
    Z = fe.forward_rect(torch.ones(n_cols, D, device=dev), rp_d, col_d, *outs)[0]
    want = torch.from_numpy(deg.astype(np.float32)).to(dev)
    assert torch.equal(Z[:, 0], want) and torch.equal(Z[:, D - 1], want) and torch.equal(Z[:, D // 2], want)
    assert torch.equal(Z, want[:, None].expand(N, D))
    del Z
    m = int(min(1021, (2 ** 24) // max(int(deg.max()), 1)))
    assert m >= 2
    ids = torch.arange(n_cols, device=dev, dtype=torch.float32) % m
    X = ids[:, None].expand(n_cols, D).contiguous()
    Z = fe.forward_rect(X, rp_d, col_d, *outs)[0]
    cs = np.concatenate([[0.0], np.cumsum((col.astype(np.int64) % m).astype(np.float64))])
    want = torch.from_numpy((cs[rp[1:]] - cs[rp[:-1]]).astype(np.float32)).to(dev)
    ht = outs[3]
    dense_rows = torch.repeat_interleave(ht != 0, 16)[:N]
    for sel, what in ((dense_rows, "dense-tile rows"), (~dense_rows, "sparse-row rows")):
        assert torch.equal(Z[sel][:, 0], want[sel]) and torch.equal(Z[sel][:, D - 1], want[sel]), what
    del X, Z
    X1 = torch.randn(n_cols, D, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    Z1 = fe.forward_rect(X1, rp_d, col_d, *outs)[0]
    assert torch.equal(Z1, fe.forward_rect(X1, rp_d, col_d, *outs)[0])
    # sampled rows vs the oracle: a sub-graph of the sampled rows over only the X rows they reference
    rng = np.random.default_rng(0)
    rows = np.sort(rng.choice(N, 2000, replace=False))
    hubs = np.argsort(deg)[-8:]  # plus the heaviest rows (split into segments + fix-up)
    dense_w = np.nonzero(ht.cpu().numpy())[0][:8]
    rows = np.unique(np.concatenate([rows, hubs] + [np.arange(16 * w, min(16 * w + 16, N)) for w in dense_w]))
    sub_col = np.concatenate([col[rp[r]:rp[r + 1]] for r in rows]).astype(np.int64)
    used, inv = np.unique(sub_col, return_inverse=True)
    sub_rp = np.concatenate([[0], np.cumsum(deg[rows])]).astype(np.int32)
    X_sub = X1[torch.from_numpy(used).to(dev)].cpu().numpy()
    ok, ratio = oracle_mod.check_spmm(Z1[torch.from_numpy(rows).to(dev)].cpu().numpy(), sub_rp, inv.astype(np.int32), X_sub)
    assert ok, ratio
    del X1, Z1
    torch.cuda.empty_cache()
    return h


def test_config4_products_scale_full_size(oracle_mod, dev):
    """BASELINE config 4 at FULL size on one GPU: 2.45 M nodes / 62 M stored entries, dim 256 (X and Z 2.5 GB each)."""
    rp, col = graphs.powerlaw_graph(2450000, 62000000, seed=4)
    assert len(rp) - 1 == 2450000 and abs(len(col) - 62000000) < 62000
    h = _full_size_properties(oracle_mod, dev, rp, col, 2450000, 256)
    assert h.n_split_rows > 0 and h.n_tasks + h.n_slice_tasks > 2000000 and h.n_slices == 8


def test_config5_share_dense_heavy(oracle_mod, dev):
    """BASELINE config 5 as SURVEY.md 8(d) defines it, one GPU's share: 2 M rows x 16 M columns, 32 M entries, planted
    16-row groups sharing <= 24 columns -- a majority of windows on the dense-tile path under the reference's
    classifier (rule 0) -- all 16 M rows of X resident (8.2 GB, N*D = 2.05e9 > 2^31)."""
    rp, col = graphs.planted_powerlaw_block(2000000, 16000000, 32000000, seed=3)
    assert len(col) == 32000000 and int(col.max()) > 15_900_000
    h = _full_size_properties(oracle_mod, dev, rp, col, 16000000, 128)
    W = 125000
    assert h.n_dense > 0.6 * W and h.max_dense_k <= 24 and h.n_dense_compact == h.n_dense
    assert h.nnz_dense > 0.4 * len(col) and h.nnz_sparse > 0.4 * len(col)  # both sub-paths carry real work


def test_config5_full_size_on_one_gpu(oracle_mod, dev):
    """BASELINE config 5 at FULL size on ONE GPU (it fits 288 GB): 16 M nodes / 256 M stored entries, dim 128, 70 % of the
    16-row windows planted groups of 8-24 columns (dense-tile path under the reference's classifier) -- the graph
    `bench.py --workload c5` times (bench.make_strong_block("c5", 1, 0): 64 row chunks, generated here by 16 spawned host
    processes).  X and Z are 8.2 GB each, N*D = 2.05e9 > 2^31.  Same size-independent properties as the other full-size tests."""
    import bench
    rp, col, n_local, n_total = bench.make_strong_block("c5", 1, 0, workers=16)
    assert n_local == n_total == 16000000 and abs(len(col) - 256000000) < 256000 and int(col.max()) > 15_990_000
    h = _full_size_properties(oracle_mod, dev, rp, col, n_total, 128)
    W = 1000000
    assert h.n_dense > 0.6 * W and h.max_dense_k <= 24 and h.n_dense_compact == h.n_dense
    assert h.nnz_dense > 0.4 * len(col) and h.nnz_sparse > 0.4 * len(col) and h.n_slices == 8
