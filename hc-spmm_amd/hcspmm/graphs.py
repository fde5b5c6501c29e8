"""Synthetic graphs for tests and bench.py (there is no dataset in the image: the reference's
Dataset.zip is missing).  All generators are seeded, return int32 CSR arrays
(row_pointers[N+1], column_index[E]) with columns ascending and unique per row -- the invariants
scipy's coo->csr gives the reference (dataset.py:95-96) -- and have no self loops.
"""
import numpy as np
import scipy.sparse as sp


def _to_csr(rows, cols, N):
    """Binary CSR from (possibly duplicated) coordinate lists; duplicates merged like scipy tocsr()."""
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    m = sp.coo_matrix((np.ones(rows.shape[0], np.int8), (rows, cols)), shape=(N, N)).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32)


def powerlaw_graph(num_nodes, num_edges, seed=0, exponent=2.1, symmetric=True, shuffle=True, max_degree_frac=0.02):
    """Chung-Lu style power-law graph with ~num_edges stored entries (directed count).

    Endpoint weights w_i ~ (i + i0)^(-1/(exponent-1)); hubs are capped at max_degree_frac*N
    expected degree; vertex ids are shuffled so hubs are spread over the row windows.
    """
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    alpha = 1.0 / (exponent - 1.0)
    w = (np.arange(N, dtype=np.float64) + 1.0) ** (-alpha)
    w /= w.sum()
    if max_degree_frac is not None:
        cap = max_degree_frac * N / max(num_edges, 1)
        for _ in range(8):
            w = np.minimum(w, cap)
            w /= w.sum()
    cdf = np.cumsum(w)
    cdf[-1] = 1.0
    target = int(num_edges)
    pairs = target // 2 if symmetric else target
    have_r = np.empty(0, np.int64)
    have_c = np.empty(0, np.int64)
    perm = rng.permutation(N) if shuffle else np.arange(N)
    for _ in range(6):
        need = pairs - (have_r.shape[0] // (2 if symmetric else 1))
        if need <= 0:
            break
        draw = int(need * 1.15) + 16
        a = np.searchsorted(cdf, rng.random(draw)).astype(np.int64)
        b = np.searchsorted(cdf, rng.random(draw)).astype(np.int64)
        ok = a != b
        a, b = perm[a[ok]], perm[b[ok]]
        if symmetric:
            r = np.concatenate([have_r, a, b])
            c = np.concatenate([have_c, b, a])
        else:
            r = np.concatenate([have_r, a])
            c = np.concatenate([have_c, b])
        key = np.unique(r * N + c)
        have_r, have_c = key // N, key % N
    if symmetric:
        # trim symmetric pairs to the target (keep both directions of a pair together)
        up = have_r < have_c
        ur, uc = have_r[up], have_c[up]
        if ur.shape[0] > pairs:
            sel = rng.choice(ur.shape[0], pairs, replace=False)
            ur, uc = ur[sel], uc[sel]
        have_r, have_c = np.concatenate([ur, uc]), np.concatenate([uc, ur])
    elif have_r.shape[0] > target:
        sel = rng.choice(have_r.shape[0], target, replace=False)
        have_r, have_c = have_r[sel], have_c[sel]
    return _to_csr(have_r, have_c, N)


def uniform_graph(num_nodes, num_edges, seed=0):
    """Erdos-Renyi style directed graph with ~num_edges entries."""
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    r = rng.integers(0, N, int(num_edges * 1.05) + 8)
    c = rng.integers(0, N, r.shape[0])
    key = np.unique(r.astype(np.int64) * N + c)
    key = key[(key // N) != (key % N)]
    if key.shape[0] > num_edges:
        key = rng.choice(key, int(num_edges), replace=False)
    return _to_csr(key // N, key % N, N)


def planted_dense_graph(num_nodes, seed=0, dense_fraction=0.7, cols_per_window=(8, 24), fill=0.5, sparse_degree=12):
    """Graph whose 16-row windows mostly share a small column set (the layout LOI produces), so the
    classifier sends a large share of windows to the dense-tile path (BASELINE config 5 analogue).

    A `dense_fraction` of the windows draw K in cols_per_window shared columns and each row links
    to each of them with probability `fill`; the other windows get `sparse_degree` random columns
    per row (sparse-row path).
    """
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    W = (N + 15) // 16
    rows, cols = [], []
    is_dense = rng.random(W) < dense_fraction
    for w in range(W):
        r0, r1 = w * 16, min(w * 16 + 16, N)
        nr = r1 - r0
        if is_dense[w]:
            K = int(rng.integers(cols_per_window[0], cols_per_window[1] + 1))
            cset = rng.choice(N, K, replace=False)
            mask = rng.random((nr, K)) < fill
            rr, kk = np.nonzero(mask)
            rows.append(r0 + rr)
            cols.append(cset[kk])
        else:
            rr = np.repeat(np.arange(r0, r1), sparse_degree)
            rows.append(rr)
            cols.append(rng.integers(0, N, rr.shape[0]))
    return _to_csr(np.concatenate(rows).astype(np.int64), np.concatenate(cols).astype(np.int64), N)


def planted_dense_graph_fast(num_nodes, seed=0, dense_fraction=0.7, k_cols=16, fill=0.5, sparse_degree=12):
    """Vectorised variant of planted_dense_graph for multi-million-node benchmarks (fixed K)."""
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    W = (N + 15) // 16
    is_dense = rng.random(W) < dense_fraction
    dw = np.nonzero(is_dense)[0]
    sw = np.nonzero(~is_dense)[0]
    cset = rng.integers(0, N, (dw.shape[0], k_cols))                 # shared columns per dense window
    m = rng.random((dw.shape[0], 16, k_cols)) < fill
    wi, ri, ki = np.nonzero(m)
    drow = dw[wi] * 16 + ri
    dcol = cset[wi, ki]
    ok = drow < N
    srow = np.repeat(sw * 16, 16 * sparse_degree) + np.tile(np.repeat(np.arange(16), sparse_degree), sw.shape[0])
    ok2 = srow < N
    scol = rng.integers(0, N, srow.shape[0])
    rows = np.concatenate([drow[ok], srow[ok2]]).astype(np.int64)
    cols = np.concatenate([dcol[ok], scol[ok2]]).astype(np.int64)
    return _to_csr(rows, cols, N)


_CDF_CACHE = {}


def _capped_powerlaw_cdf(n, cap_prob, exponent=2.1):
    """CDF of endpoint weights w_i ~ (i+1)^(-1/(exponent-1)) with every probability capped at cap_prob.  (The last result
    is kept: the chunks of one strong-scaling graph -- bench.py make_strong_block -- ask for the same 16 M-entry table.)"""
    key = (int(n), float(cap_prob), float(exponent))
    if key in _CDF_CACHE:
        return _CDF_CACHE[key]
    c = _capped_powerlaw_cdf_uncached(n, cap_prob, exponent)
    if len(_CDF_CACHE) >= 4:
        _CDF_CACHE.clear()
    _CDF_CACHE[key] = c
    return c


def _column_relabelling(n_cols, seed):
    key = ("perm", int(n_cols), int(seed))
    if key not in _CDF_CACHE:
        if len(_CDF_CACHE) >= 4:
            _CDF_CACHE.clear()
        _CDF_CACHE[key] = np.random.default_rng(seed).permutation(n_cols)
    return _CDF_CACHE[key]


def _capped_powerlaw_cdf_uncached(n, cap_prob, exponent=2.1):
    alpha = 1.0 / (exponent - 1.0)
    w = (np.arange(n, dtype=np.float64) + 1.0) ** (-alpha)
    w /= w.sum()
    for _ in range(8):
        w = np.minimum(w, cap_prob)
        w /= w.sum()
    c = np.cumsum(w)
    c[-1] = 1.0
    return c


def powerlaw_block(n_rows, n_cols, n_entries, seed=0, rank=0, rows=None):
    """Row block of a power-law graph: `n_rows` local rows, column ids global in [0, n_cols), ~n_entries stored
    entries.  Row degrees follow a (shuffled) local power law, columns a global one whose relabelling depends on
    `seed` only -- so every rank of a sharded job agrees on which columns are popular -- while the draws depend on
    (seed, rank): cheap to generate per rank, no exchange.  `rows` (optional, ascending local ids): put the
    entries on these rows only.  Returns (row_pointers[n_rows+1], column_index) int32."""
    rng = np.random.default_rng(seed + 1000 * rank)
    n_draw_rows = int(n_rows if rows is None else len(rows))
    if n_draw_rows == 0 or n_entries <= 0:
        return np.zeros(n_rows + 1, np.int32), np.zeros(0, np.int32)
    cap = 0.02 * max(n_draw_rows, 1) / max(n_entries, 1)
    rcdf = _capped_powerlaw_cdf(n_draw_rows, cap)
    ccdf = _capped_powerlaw_cdf(n_cols, cap)
    rperm = rng.permutation(n_draw_rows)
    cperm = _column_relabelling(n_cols, seed)  # same column relabelling on every rank
    draw = int(n_entries * 1.12)
    r = rperm[np.searchsorted(rcdf, rng.random(draw))].astype(np.int64)
    if rows is not None:
        r = np.asarray(rows, dtype=np.int64)[r]
    c = cperm[np.searchsorted(ccdf, rng.random(draw))].astype(np.int64)
    key = np.unique(r * n_cols + c)
    if key.shape[0] > n_entries:
        key = np.sort(rng.choice(key, int(n_entries), replace=False))
    rr, cc = key // n_cols, key % n_cols
    rp = np.zeros(n_rows + 1, np.int64)
    np.add.at(rp, rr + 1, 1)
    return np.cumsum(rp).astype(np.int32), cc.astype(np.int32)


def planted_powerlaw_block(n_rows, n_cols, n_entries, seed=0, rank=0, dense_fraction=0.7, k_range=(8, 24), fill=0.7,
                           chunk_windows=1 << 16):
    """BASELINE config 5's mix as SURVEY.md 8(d) defines it ("planted 16-row groups sharing <= 24 columns so that
    the classifier sends a large fraction of windows to the dense path"), as a row block: n_rows local rows,
    column ids global in [0, n_cols), ~n_entries stored entries (n_cols == n_rows: the whole square graph).

    A `dense_fraction` of the 16-row windows are planted groups -- the layout a LOI reorder produces: K in k_range
    shared columns (uniform over ALL n_cols, i.e. no locality in X), each row linked to each of them with
    probability `fill`.  With K <= 24 and fill >= 0.25 the reference's classifier (rule 0: 0.1985*(K-1) -
    6.578*density - 3.149 <= 0) sends every such window to the dense-tile path.  The other windows are
    unstructured power-law rows (sparse-row path) holding the rest of the entries.  Duplicate-free, columns
    ascending per row; generated in chunks so the 16 M-node full size stays within a few GB of host memory."""
    rng = np.random.default_rng(seed + 7919 * rank + 1)
    N, M = int(n_rows), int(n_cols)
    W = (N + 15) // 16
    kmax = int(k_range[1])
    is_dense = rng.random(W) < dense_fraction
    dw = np.nonzero(is_dense)[0]
    r_parts, c_parts = [], []
    n_planted = 0
    for a in range(0, dw.shape[0], chunk_windows):
        wch = dw[a:a + chunk_windows]
        K = rng.integers(k_range[0], kmax + 1, wch.shape[0])
        cset = rng.integers(0, M, (wch.shape[0], kmax), dtype=np.int64)
        m = (rng.random((wch.shape[0], 16, kmax), dtype=np.float32) < fill) & (np.arange(kmax)[None, None, :] < K[:, None, None])
        wi, ri, ki = np.nonzero(m)
        rr = wch[wi] * 16 + ri
        ok = rr < N
        r_parts.append(rr[ok].astype(np.int64))
        c_parts.append(cset[wi, ki][ok])
        n_planted += int(ok.sum())
    srows = np.nonzero(~np.repeat(is_dense, 16)[:N])[0]
    rp_s, col_s = powerlaw_block(N, M, max(int(n_entries) - n_planted, 0), seed=seed, rank=rank, rows=srows)
    r_parts.append(np.repeat(np.arange(N, dtype=np.int64), np.diff(rp_s)))
    c_parts.append(col_s.astype(np.int64))
    key = np.unique(np.concatenate(r_parts) * M + np.concatenate(c_parts))
    del r_parts, c_parts
    rr, cc = key // M, key % M
    rp = np.zeros(N + 1, np.int64)
    np.add.at(rp, rr + 1, 1)
    return np.cumsum(rp).astype(np.int32), cc.astype(np.int32)


def community_graph(num_nodes, num_edges, seed=0, group_rows=(8, 40), pool_cols=(4, 12), links_per_row=(1, 3), shuffle=True):
    """Community-structured low-degree graph of the paper's RD order (Table II: 4.86 M nodes / 10.1 M entries) whose
    structure is hidden by the vertex numbering -- the input the LOI reorder (LOI.cpp:660-805) exists for.

    The vertices fall into consecutive groups of `group_rows` rows.  Each group has a column pool of `pool_cols`
    of its OWN members, and every row of the group links to `links_per_row` distinct pool columns: rows of a group
    share columns (what LOI groups on, what the dense-tile path's 16-row reuse feeds on), and the columns a group
    reads lie together once the group does (what the L2 feeds on).  The remaining entries, up to `num_edges`, are
    power-law noise: uniform rows, columns from a capped power law (hub columns).  With `shuffle` the vertex ids
    are then permuted uniformly (rows and columns alike), which leaves an isomorphic graph with no 16-row window
    structure at all.  Returns (row_pointers, column_index, group_of_vertex) -- the last in the FINAL numbering, so a
    reorder can be scored against the planted truth."""
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    sizes = rng.integers(group_rows[0], group_rows[1] + 1, N // group_rows[0] + 2)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    starts = starts[starts < N]
    G = starts.shape[0]
    ends = np.append(starts[1:], N)
    gsize = ends - starts
    idx = np.arange(N, dtype=np.int64)
    grp = np.searchsorted(starts, idx, side="right") - 1
    K = np.minimum(rng.integers(pool_cols[0], pool_cols[1] + 1, G), gsize)          # pool size per group
    pool_off = rng.integers(0, 1 << 30, G) % np.maximum(gsize - K + 1, 1)           # pool = K consecutive members
    lmax = int(links_per_row[1])
    m = rng.integers(links_per_row[0], lmax + 1, N)                                  # links per row
    # `lmax` distinct pool slots per row: a random start and stride 1 inside the pool (distinct while m <= K)
    first = rng.integers(0, 1 << 30, N) % K[grp]
    j = np.arange(lmax, dtype=np.int64)[None, :]
    slot = (first[:, None] + j) % K[grp][:, None]
    ok = (j < m[:, None]) & (j < K[grp][:, None])
    rows_p = np.broadcast_to(idx[:, None], slot.shape)[ok]
    cols_p = (starts[grp] + pool_off[grp])[:, None] + slot
    cols_p = cols_p[ok]
    n_noise = max(int(num_edges) - rows_p.shape[0], 0)
    cdf = _capped_powerlaw_cdf(N, 0.02 * N / max(int(num_edges), 1))
    cperm = rng.permutation(N)
    draw = int(n_noise * 1.02) + 8
    rows_n = rng.integers(0, N, draw)
    cols_n = cperm[np.searchsorted(cdf, rng.random(draw))]
    rows = np.concatenate([rows_p, rows_n]).astype(np.int64)
    cols = np.concatenate([cols_p, cols_n]).astype(np.int64)
    if shuffle:
        relabel = rng.permutation(N)
        rows, cols = relabel[rows], relabel[cols]
        group_of = np.empty(N, np.int32)
        group_of[relabel] = grp.astype(np.int32)
    else:
        group_of = grp.astype(np.int32)
    rp, col = _to_csr(rows, cols, N)
    return rp, col, group_of


def molecule_graph(num_nodes, seed=0, size_range=(10, 46), heavy_fraction=0.45):
    """Collection of small molecule-like components laid out one after another (the shape of the
    TU-collection datasets in the paper's Table II -- YeastH, OVCAR-8H, ...: millions of nodes, average
    degree ~2, every neighbour within a few dozen ids): "heavy atoms" form a branched chain with a
    few ring closures, the remaining nodes are leaves of the latest heavy atom.  Symmetric."""
    rng = np.random.default_rng(seed)
    N = int(num_nodes)
    sizes = rng.integers(size_range[0], size_range[1], N // size_range[0] + 2)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    starts = starts[starts < N]
    idx = np.arange(N, dtype=np.int64)
    comp_start = starts[np.searchsorted(starts, idx, side="right") - 1]
    heavy = rng.random(N) < heavy_fraction
    heavy[starts] = True

    def last_heavy_at_or_before(i):  # i: int64 array of node ids (>= 0)
        return np.maximum.accumulate(np.where(heavy, idx, -1))[i]
    prev1 = last_heavy_at_or_before(np.maximum(idx - 1, 0))              # latest heavy atom before i
    prev2 = last_heavy_at_or_before(np.maximum(prev1 - 1, 0))            # the one before that
    prev2 = np.where(prev2 >= comp_start, prev2, prev1)
    parent = np.where(heavy & (rng.random(N) < 0.25), prev2, prev1)
    has_parent = idx > comp_start
    ring_to = last_heavy_at_or_before(np.maximum(idx - 9, 0))
    ring = heavy & (rng.random(N) < 0.12) & (ring_to >= comp_start) & (idx - 9 >= comp_start)
    a = np.concatenate([idx[has_parent], idx[ring]])
    b = np.concatenate([parent[has_parent], ring_to[ring]])
    return _to_csr(np.concatenate([a, b]), np.concatenate([b, a]), N)


def write_coo_text(path, rowptr, col):
    """Reference on-disk format: one "dst,src" line per entry, 1-based, sorted by src
    (dataset.py:52-53 reads it; LOI.cpp:493-499 needs the second field ascending)."""
    with open(path, "w") as f:
        for r in range(len(rowptr) - 1):
            for e in range(rowptr[r], rowptr[r + 1]):
                f.write("%d,%d\n" % (col[e] + 1, r + 1))
