"""CPU tests of the PRODUCT's host side (libhcspmm.so through its C ABI): exported symbols,
preprocess / plan / LOI integer parity against the oracle and the reference-generated golden
vectors, argument checking.  No device compute is called here."""
import ctypes
import glob
import os
import re
import sys

import numpy as np
import pytest
import torch

import hcspmm
from hcspmm import graphs
from hcspmm.capi import Header, PlanParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _pre(rp, col, rule=0):
    N = len(rp) - 1
    return hcspmm.preprocess(_t(col), _t(rp), N, len(col), (N + 15) // 16, rule=rule)


def test_library_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "hcspmm.h")).read()
    declared = set(re.findall(r"\b(hcspmm_[a-z0-9_]+)\s*\(", header))
    declared -= {"hcspmm_plan_header", "hcspmm_plan_params"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.lib()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.hcspmm_abi_version() == 3
    assert ctypes.sizeof(Header) == 4 * Header.WORDS


def test_preprocess_rule_by_embedding_width(oracle_mod):
    rp, col = graphs.planted_dense_graph(640, seed=22)
    N = len(rp) - 1
    args = (_t(col), _t(rp), N, len(col), (N + 15) // 16)
    for dim, rule in ((32, 3), (63, 3), (64, 4), (256, 4)):
        got = hcspmm.preprocess(*args, rule="mi355x", dim=dim)
        assert np.array_equal(got[3].numpy(), oracle_mod.preprocess(rp, col, rule)[3])
    with pytest.raises(RuntimeError, match="needs dim"):
        hcspmm.preprocess(*args, rule="mi355x")


def test_forward_argument_checks_need_no_gpu(capi):
    """Bad arguments are refused before anything touches the device (so this runs on a CPU-only box)."""
    L = capi.lib()
    z = ctypes.c_void_p(0)
    one = np.zeros(4, np.int32)
    p = ctypes.c_void_p(one.ctypes.data)
    common = (z, z, z, z, z, z, z, None)  # graph arrays, plan, header
    assert L.hcspmm_forward_typed(p, 1, 4, p, 4, 7, *common, 1, 0, 4, z, 0, z) == capi.EINVAL      # unknown dtype
    assert L.hcspmm_forward_typed(p, 1, 4, p, 4, 0, *common, -1, 0, 4, z, 0, z) == capi.EINVAL     # negative N
    assert L.hcspmm_forward_typed(p, 1, 2, p, 4, 0, *common, 1, 0, 4, z, 0, z) == capi.EINVAL      # ldx < D
    assert L.hcspmm_forward_typed(p, 1, 4, p, 4, 2, *common, 0, 0, 4, z, 0, z) == 0                # N == 0: nothing to do
    assert L.hcspmm_forward_typed(z, 1, 4, p, 4, 1, *common, 1, 0, 4, z, 0, z) == capi.EINVAL      # null X
    assert L.hcspmm_wide_threshold_typed(None, 128, 9) == 2**31 - 1
    assert L.hcspmm_wide_threshold_typed(None, 32, 0) == 64 and L.hcspmm_wide_threshold_typed(None, 256, 0) == 2**31 - 1
    assert L.hcspmm_wide_threshold_typed(None, 256, 2) == 64  # 16-bit rows of 256 columns still fit 32 lanes


def test_missing_library_fails_loudly(monkeypatch, capi):
    monkeypatch.setattr(capi, "_LIB", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libhcspmm.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        capi.lib()


GRAPHS = [
    ("powerlaw_10k", lambda: graphs.powerlaw_graph(10000, 50000, seed=1)),       # BASELINE config 2 shape
    ("powerlaw_ragged", lambda: graphs.powerlaw_graph(1003, 20000, seed=2)),      # N % 16 != 0, hubs
    ("uniform", lambda: graphs.uniform_graph(2048, 30000, seed=3)),
    ("planted", lambda: graphs.planted_dense_graph(1500, seed=4)),
    ("empty", lambda: (np.zeros(50, np.int32), np.zeros(0, np.int32))),
    ("single", lambda: (np.array([0, 1], np.int32), np.array([0], np.int32))),
]


@pytest.mark.parametrize("name,gen", GRAPHS, ids=[g[0] for g in GRAPHS])
@pytest.mark.parametrize("rule", [0, 1, 2, 3, 4])
def test_preprocess_bit_exact_vs_oracle(oracle_mod, name, gen, rule):
    rp, col = gen()
    want = oracle_mod.preprocess(rp, col, rule)
    got = _pre(rp, col, rule)
    for w, g, n in zip(want, got[:4], ("blockPartition", "edgeToColumn", "edgeToRow", "hybrid_type")):
        assert g.dtype == torch.int32
        assert np.array_equal(w, g.numpy()), n
    assert got[5].tolist() == [0]  # col_nzr stays the reference's placeholder (hybrid_all_kernel.cu:405)


def test_preprocess_handles_rows_that_are_not_ascending(oracle_mod):
    """dataset.py hands over ascending rows, and the window pass exploits that (a 16-way merge instead of a sort); the
    reference's sort + binary search copes with any order, and so must we: same integers as the oracle on rows whose
    entries were shuffled (every window then takes the sorting fallback for its column list)."""
    rp, col = graphs.planted_dense_graph(900, seed=31)
    rng = np.random.default_rng(1)
    col = col.copy()
    for r in rng.choice(len(rp) - 1, 300, replace=False):
        seg = col[rp[r]:rp[r + 1]]
        rng.shuffle(seg)
    for rule in (0, 2):
        want = oracle_mod.preprocess(rp, col, rule)
        got = _pre(rp, col, rule)
        for w, g in zip(want, got[:4]):
            assert np.array_equal(w, g.numpy())


def test_preprocess_multithreaded_equals_single(capi):
    rp, col = graphs.powerlaw_graph(60000, 400000, seed=5)
    N, E = len(rp) - 1, len(col)
    W = (N + 15) // 16
    outs = []
    for threads in (1, 7):
        bp, ht = np.zeros(W, np.int32), np.zeros(W, np.int32)
        e2c, e2r = np.zeros(E, np.int32), np.zeros(E, np.int32)
        rc = capi.lib().hcspmm_preprocess_host(rp.ctypes.data, col.ctypes.data, N, E, N, 0, threads, bp.ctypes.data,
                                               e2c.ctypes.data, e2r.ctypes.data, ht.ctypes.data)
        assert rc == 0
        outs.append((bp, e2c, e2r, ht))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_preprocess_rejects_bad_arguments(capi):
    L = capi.lib()
    rp = np.array([0, 2, 1], np.int32)  # not monotone
    col = np.array([0], np.int32)
    out = np.zeros(4, np.int32)
    assert L.hcspmm_preprocess_host(rp.ctypes.data, col.ctypes.data, 2, 1, 0, 0, 1, out.ctypes.data, out.ctypes.data,
                                    out.ctypes.data, out.ctypes.data) == capi.EINVAL
    assert L.hcspmm_preprocess_host(None, None, 2, 1, 0, 0, 1, None, None, None, None) == capi.EINVAL
    rp = np.array([0, 1], np.int32)
    assert L.hcspmm_preprocess_host(rp.ctypes.data, col.ctypes.data, 1, 1, 0, 9, 1, out.ctypes.data, out.ctypes.data,
                                    out.ctypes.data, out.ctypes.data) == capi.EINVAL  # unknown rule
    with pytest.raises(RuntimeError, match="num_row_windows"):
        hcspmm.preprocess(_t(col), _t(rp), 1, 1, 5)


def _decode_plan(plan):
    h = Header.from_buffer_copy(plan[:Header.WORDS].tobytes())
    tasks = plan[h.off_tasks:h.off_tasks + 4 * h.n_tasks].reshape(-1, 4)
    dindex = plan[h.off_dense_index:h.off_dense_index + 4 * h.n_dense].reshape(-1, 4)
    fix = plan[h.off_fixups:h.off_fixups + 4 * h.n_split_rows].reshape(-1, 4)
    return h, tasks, dindex, fix


def _decode_slices(plan, h):
    """XCD-affine column slices: -> (table, [descriptors of slice s without the padding]) after checking the layout
    (offsets multiples of 64, padding = (-1, 0, 0, -1) at the end of every list, longest length class first, rows
    ascending inside a class)."""
    if h.n_slices == 0:
        assert h.n_slice_tasks == 0 and h.slice_xcd_tasks == 0 and h.nnz_sliced == 0
        return np.zeros(1, np.int64), []
    table = plan[h.off_slice_table:h.off_slice_table + h.n_slices + 1].astype(np.int64)
    assert table[0] == 0 and table[-1] == h.n_slice_tasks and np.all(np.diff(table) >= 0) and np.all(table % 64 == 0)
    per_xcd = [int(sum(table[s + 1] - table[s] for s in range(x, h.n_slices, 8))) for x in range(8)]
    assert h.slice_xcd_tasks == max(per_xcd)
    desc = plan[h.off_slice_tasks:h.off_slice_tasks + 4 * h.n_slice_tasks].reshape(-1, 4)
    out = []
    for s in range(h.n_slices):
        d = desc[table[s]:table[s + 1]]
        real = d[:, 0] >= 0
        n = int(real.sum())
        assert np.all(real[:n]) and np.all(d[n:] == np.array([-1, 0, 0, -1])) and len(d) - n < 64
        d = d[:n]
        lens = d[:, 2]
        assert np.all(lens >= 1) and np.all(lens <= h.segment_len)
        cls = np.ceil(np.log2(np.maximum(lens, 1))).astype(int) + 1
        assert np.all(np.diff(cls) <= 0)
        for c in np.unique(cls):
            assert np.all(np.diff(d[cls == c, 0]) >= 0)
        out.append(d)
    return table, out


def _expand_tiny(h, tasks, fix, rp, col):
    """The last n_tiny descriptors are (row | -(slot+1), index0, length, index1): turn them back into
    (row, e0, length, slot) after checking the inline indices against the CSR arrays."""
    tasks = tasks.copy()
    slot_row = {}
    for row, s0, ns, _ in fix:
        for s in range(s0, s0 + ns):
            slot_row[s] = row
    first = h.n_tasks - h.n_tiny
    assert np.all(tasks[:first, 2] > 2) and np.all(tasks[first:, 2] <= 2)
    for i in range(first, h.n_tasks):
        x, i0, ln, i1 = tasks[i]
        slot = -1 if x >= 0 else -(x + 1)
        row = x if x >= 0 else slot_row[slot]
        seg = col[rp[row]:rp[row + 1]]
        if ln == 0:
            assert (i0, i1) == (-1, -1) and slot < 0
            e0 = rp[row]
        else:
            e0 = rp[row] + int(np.searchsorted(seg, i0))
            assert col[e0] == i0 and (i1 == (col[e0 + 1] if ln == 2 else -1))
        tasks[i] = (row, e0, ln, slot)
    return tasks


def _threaded_plan_graph():
    """Enough windows (>= 4096) for the multi-threaded plan build: planted dense windows, low-degree rows, and
    hub rows longer than the split threshold."""
    rp, col = graphs.planted_dense_graph_fast(70000, seed=8, dense_fraction=0.4, k_cols=12, fill=0.5, sparse_degree=3)
    N = len(rp) - 1
    rng = np.random.default_rng(8)
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
    hubs = np.array([17, 30011, 69990])
    extra_r = np.repeat(hubs, [700, 1300, 520])
    extra_c = rng.integers(0, N, extra_r.shape[0])
    return graphs._to_csr(np.concatenate([rows, extra_r]), np.concatenate([col.astype(np.int64), extra_c]), N)


_PLAN_GRAPHS = GRAPHS[:4] + [("threaded_70k", _threaded_plan_graph)]


@pytest.mark.parametrize("name,gen", _PLAN_GRAPHS, ids=[g[0] for g in _PLAN_GRAPHS])
def test_plan_covers_every_entry_exactly_once(name, gen):
    rp, col = gen()
    N, E = len(rp) - 1, len(col)
    bp, e2c, e2r, ht, plan_t, _ = _pre(rp, col, 0)
    plan = plan_t.numpy()
    h, tasks, dindex, fix = _decode_plan(plan)
    tasks = _expand_tiny(h, tasks, fix, rp, col)
    assert h.magic == Header.MAGIC and h.num_nodes == N and h.num_edges == E and h.total_words <= len(plan)
    ht = ht.numpy()
    cover = np.zeros(E, np.int32)
    rows_written = np.zeros(N, np.int32)
    # sparse tasks: sorted by descending power-of-two length class (0, 1, 2, 3-4, 5-8, ...), rows
    # ascending inside a class; each <= split_threshold
    lens = tasks[:, 2]
    cls = np.where(lens > 0, np.ceil(np.log2(np.maximum(lens, 1))).astype(int) + 1, 0)
    assert np.all(np.diff(cls) <= 0) and (len(lens) == 0 or lens.max() <= h.split_threshold)
    for c in np.unique(cls):
        assert np.all(np.diff(tasks[cls == c, 0]) >= 0)  # row order inside a class
    for row, e0, ln, slot in tasks:
        assert ht[row // 16] == 0 or rp[min((row // 16) * 16 + 16, N)] == rp[(row // 16) * 16]
        assert rp[row] <= e0 and e0 + ln <= rp[row + 1]
        cover[e0:e0 + ln] += 1
        if slot < 0:
            assert e0 == rp[row] and ln == rp[row + 1] - rp[row]
            rows_written[row] += 1
    # fix-ups: consecutive slots, segments tile the row in order
    slots = {}
    for row, e0, ln, slot in tasks:
        if slot >= 0:
            slots[slot] = (row, e0, ln)
    assert sorted(slots) == list(range(h.n_partials))
    for row, s0, ns, _ in fix:
        e = rp[row]
        for s in range(s0, s0 + ns):
            assert slots[s][0] == row and slots[s][1] == e
            e += slots[s][2]
        assert e == rp[row + 1] and rp[row + 1] - rp[row] > h.split_threshold
        rows_written[row] += 1
    # dense windows: U reproduces the sorted unique columns, masks reproduce the 0/1 tiles
    pack = plan[h.off_dense_pack:]
    compact = plan[h.off_dense_compact:]
    CK, CK2 = 40, 80  # HCSPMM_COMPACT_K, HCSPMM_COMPACT2_K
    compact2 = plan[h.off_dense_compact2:]
    assert h.off_dense_compact % 64 == 0 and h.n_dense_compact == int((dindex[:, 2] <= CK // 4).sum())
    assert h.off_dense_compact2 % 64 == 0 and h.off_dense_compact == h.off_dense_compact2 + 128 * h.n_dense_compact2
    assert h.n_dense_compact2 == int(((dindex[:, 2] > CK // 4) & (dindex[:, 2] <= CK2 // 4)).sum())
    assert np.all(np.diff(dindex[:, 2]) <= 0)  # widest first: regular, double-record (K <= 80), compact (K <= 40)
    for w, off, K4, is_compact in dindex:
        assert ht[w] == 1
        K = 4 * K4
        assert K == 8 * bp.numpy()[w] and is_compact == (1 if K <= CK else (2 if K <= CK2 else 0))
        lo, hi = rp[w * 16], rp[min(w * 16 + 16, N)]
        uniq = np.unique(col[lo:hi])
        if is_compact:  # fixed 64- / 128-word record: window, K/4, U[40 / 80], 10 / 20 x (mask lo, mask hi), pad
            words, kmax, sec = (64, CK, compact) if is_compact == 1 else (128, CK2, compact2)
            rec = sec[off:off + words]
            m0 = 2 + kmax
            assert off % words == 0 and rec[0] == w and rec[1] == K4 and np.all(rec[2 + K:m0] == -1)
            U, masks = rec[2:2 + K], rec[m0:m0 + 2 * K4].view(np.uint64)
            assert np.all(rec[m0 + 2 * K4:] == 0)
        else:
            U, masks = pack[off:off + K], pack[off + K:off + K + 2 * K4].view(np.uint64)
        assert np.array_equal(U[:len(uniq)], uniq) and np.all(U[len(uniq):] == -1)
        tile = np.zeros((16, K), np.int32)
        for kk in range(K4):
            for lane in range(64):
                if (int(masks[kk]) >> lane) & 1:
                    tile[lane & 15, 4 * kk + (lane >> 4)] = 1
        want = np.zeros((16, K), np.int32)
        for r in range(w * 16, min(w * 16 + 16, N)):
            want[r - w * 16, np.searchsorted(uniq, col[rp[r]:rp[r + 1]])] = 1
        assert np.array_equal(tile, want)
        cover[lo:hi] += 1
        rows_written[w * 16:min(w * 16 + 16, N)] += 1
    assert np.all(cover == 1) and np.all(rows_written == 1)
    assert list(h.n_len_gt) == [int((lens > (16 << b)).sum()) for b in range(5)]
    assert h.nnz_sparse + h.nnz_dense == E


def test_plan_splits_hub_rows(capi):
    # one row with 2000 entries, threshold 512 / segment 256 -> 8 segments
    N = 2100
    deg = np.zeros(N, np.int64)
    deg[7] = 2000
    deg[100:200] = 3
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    rng = np.random.default_rng(0)
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg if d]).astype(np.int32)
    plan = _pre(rp, col, 2)[4].numpy()
    h, tasks, _, fix = _decode_plan(plan)
    assert h.n_split_rows == 1 and h.n_partials == 8 and fix.tolist() == [[7, 0, 8, 0]]
    assert tasks[:8, 2].tolist() == [256] * 7 + [208] or sorted(tasks[:8, 2].tolist(), reverse=True) == [256] * 7 + [208]
    ws = capi.lib().hcspmm_workspace_bytes(ctypes.byref(h), 128)
    assert ws == 8 * 128 * 4


def test_plan_tiny_descriptors_carry_indices_inline():
    # rows of 0/1/2 entries and a 514-entry row whose last segment (2 entries) is a tiny descriptor too
    deg = np.array([0, 1, 2, 3, 514, 2, 0, 1] + [0] * 600, np.int64)
    N = len(deg)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    rng = np.random.default_rng(4)
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg if d]).astype(np.int32)
    plan = _pre(rp, col, 2)[4].numpy()
    h, tasks, _, fix = _decode_plan(plan)
    assert h.n_tiny == int((deg <= 2).sum()) + 1 and fix.tolist() == [[4, 0, 3, 0]]
    tiny = tasks[h.n_tasks - h.n_tiny:]
    assert tiny[:, 2].tolist() == sorted(tiny[:, 2].tolist(), reverse=True)  # classes 2, 1, 0
    seg = tiny[tiny[:, 0] < 0]
    assert seg.tolist() == [[-3, col[rp[4] + 512], 2, col[rp[4] + 513]]]      # slot 2 of the split row
    assert tiny[tiny[:, 2] == 0][:, [1, 3]].tolist() == [[-1, -1]] * int((deg == 0).sum())
    full = _expand_tiny(h, tasks, fix, rp, col)
    cover = np.zeros(len(col), np.int32)
    for row, e0, ln, slot in full:
        cover[e0:e0 + ln] += 1
    assert np.all(cover == 1)


def test_plan_check_rejects_mismatch(capi):
    rp, col = graphs.powerlaw_graph(500, 3000, seed=1)
    plan = _pre(rp, col)[4].numpy()
    h = Header.from_buffer_copy(plan[:Header.WORDS].tobytes())
    L = capi.lib()
    N, E = len(rp) - 1, len(col)
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, len(plan)) == 0
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, 0) == 0                      # buffer length unknown: not checked
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, len(plan) - 1) == capi.EPLAN  # blob longer than its buffer
    assert L.hcspmm_plan_check(ctypes.byref(h), N + 1, E, 0) == capi.EPLAN
    for field, bad in (("magic", 0), ("n_tiny", h.n_tasks + 1), ("off_fixups", h.total_words + 4),
                       ("off_sparse_windows", h.total_words), ("n_sparse_windows", h.n_sparse_windows + 1),
                       ("off_dense_index", h.off_tasks), ("nnz_sparse", h.nnz_sparse + 1), ("num_columns", 0),
                       ("total_words", h.off_sparse_windows)):
        keep = getattr(h, field)
        setattr(h, field, bad)
        assert L.hcspmm_plan_check(ctypes.byref(h), N, E, 0) == capi.EPLAN, field
        setattr(h, field, keep)
    keep = h.n_len_gt[2]
    h.n_len_gt[2] = h.n_tasks - h.n_tiny + 1  # a wide-task prefix beyond the non-tiny tasks
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, 0) == capi.EPLAN
    h.n_len_gt[2] = keep
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, 0) == 0


def test_column_ids_are_range_checked_on_the_host(capi):
    """A column id outside [0, num_columns) is refused by preprocess and by plan_build (HCSPMM_EINVAL) instead of
    becoming an out-of-bounds gather on the GPU; num_columns > num_nodes admits the row block of a sharded graph."""
    L = capi.lib()
    rp, col = graphs.powerlaw_graph(300, 2000, seed=3)
    N, E = len(rp) - 1, len(col)
    W = (N + 15) // 16
    bp, ht, e2c, e2r = (np.zeros(W, np.int32), np.zeros(W, np.int32), np.zeros(E, np.int32), np.zeros(E, np.int32))

    def pre(c, M, threads=1):
        return L.hcspmm_preprocess_host(rp.ctypes.data, c.ctypes.data, N, E, M, 0, threads, bp.ctypes.data, e2c.ctypes.data,
                                        e2r.ctypes.data, ht.ctypes.data)
    assert pre(col, N) == 0 and pre(col, 0) == 0
    for bad_value in (N, -1, 2 ** 31 - 1):
        bad = col.copy()
        bad[E // 2] = bad_value
        assert pre(bad, N) == capi.EINVAL
    wide = col.copy()
    wide[-1] = 4 * N - 1           # legal for a row block whose columns span 4*N rows of the gathered matrix
    order = np.argsort(wide[rp[N - 1]:rp[N]], kind="stable")
    wide[rp[N - 1]:rp[N]] = wide[rp[N - 1]:rp[N]][order]
    assert pre(wide, N) == capi.EINVAL and pre(wide, 4 * N) == 0
    with pytest.raises(RuntimeError, match="invalid argument"):
        hcspmm.preprocess(_t(wide), _t(rp), N, E, W)
    outs = hcspmm.preprocess(_t(wide), _t(rp), N, E, W, num_columns=4 * N)
    h = hcspmm.plan_header(outs[4])
    assert h.num_columns == 4 * N and h.n_sparse_windows + h.n_dense == W
    # plan_build on its own checks too (a plan may be built for a caller's classification)
    words = ctypes.c_int64(0)
    assert pre(col, N) == 0
    assert L.hcspmm_plan_words(rp.ctypes.data, N, E, bp.ctypes.data, ht.ctypes.data, None, ctypes.byref(words)) == 0
    plan = np.zeros(words.value, np.int32)
    bad = col.copy()
    bad[0] = N + 5
    assert L.hcspmm_plan_build(rp.ctypes.data, bad.ctypes.data, N, E, N, bp.ctypes.data, e2c.ctypes.data, ht.ctypes.data, None,
                               plan.ctypes.data, words.value) == capi.EINVAL
    assert L.hcspmm_plan_build(rp.ctypes.data, col.ctypes.data, N, E, N, bp.ctypes.data, e2c.ctypes.data, ht.ctypes.data, None,
                               plan.ctypes.data, words.value) == 0


def test_graph_fingerprint_tells_graphs_of_equal_size_apart(capi):
    L = capi.lib()
    rp, col = graphs.powerlaw_graph(400, 3000, seed=5)
    N, E = len(rp) - 1, len(col)

    def fp(r, c):
        out = ctypes.c_uint64(0)
        assert L.hcspmm_graph_fingerprint_host(r.ctypes.data, c.ctypes.data, len(r) - 1, len(c), ctypes.byref(out)) == 0
        return out.value
    base = fp(rp, col)
    assert base == fp(rp.copy(), col.copy())
    plan = _pre(rp, col)[4].numpy()
    assert Header.from_buffer_copy(plan[:Header.WORDS].tobytes()).fingerprint == base
    swapped = col.copy()          # same N, E, same multiset of ids: two entries exchanged
    i, j = rp[10], rp[200]
    swapped[i], swapped[j] = col[j], col[i]
    assert fp(rp, swapped) != base
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(torch.int32)
    rp2, col2 = hcspmm.apply_permutation(_t(rp), _t(col), perm)  # a relabelled graph keeps N and E
    assert fp(rp2.numpy(), col2.numpy()) != base
    # threaded pass (> 2^18 elements) == single-threaded definition
    rp3, col3 = graphs.powerlaw_graph(60000, 400000, seed=5)
    want = capi_fingerprint_reference(rp3, col3)
    assert fp(rp3, col3) == want


def capi_fingerprint_reference(rp, col):
    """The definition in csrc/fingerprint.h, restated with numpy (uint64 wrap-around arithmetic)."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(x):
        x = (x + np.uint64(0x9e3779b97f4a7c15)) & M
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & M
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & M
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        N, E = len(rp) - 1, len(col)
        r = np.arange(N + 1, dtype=np.uint64)
        e = np.arange(E, dtype=np.uint64)
        t_rp = mix(~((r << np.uint64(32)) | rp.astype(np.uint32).astype(np.uint64)))
        t_col = mix((e << np.uint64(32)) | col.astype(np.uint32).astype(np.uint64))
        seed = mix(np.array([(N << 32) ^ E ^ 0x4843535000000000], dtype=np.uint64))
        return int((t_rp.sum(dtype=np.uint64) + t_col.sum(dtype=np.uint64) + seed[0]) & M)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "loi_win_*.npz"))))
def test_windowed_loi_reorder_matches_reference_golden(path):
    """reorder_plus_direct / reorder_plus (LOI.cpp:286-484 / :98-284, vertex window 300) against permutations produced
    by the reference's own LOI.cpp compiled in the authoring container (tests/golden/make_loi_fixtures.py)."""
    g = np.load(path)
    for variant in ("plus_direct", "plus"):
        perm, sizes = hcspmm.loi_reorder(_t(g["row_pointers"]), _t(g["column_index"]), variant=variant)
        assert np.array_equal(sizes.numpy(), g["group_sizes_" + variant]), variant
        assert np.array_equal(perm.numpy(), g["order_" + variant]), variant
        assert sorted(perm.tolist()) == list(range(len(g["row_pointers"]) - 1))


def test_windowed_loi_reorder_refuses_inputs_outside_the_reference_domain(capi):
    """Fewer than 50 rows (the reference's `size() - 50` underflows, LOI.cpp:333), a row without entries (its window
    bound reads past the row-order array, LOI.cpp:362) or an unsorted row (its residual merge assumes ascending
    columns): HCSPMM_EINVAL instead of reproducing undefined behaviour."""
    rp, col = graphs.uniform_graph(49, 300, seed=1)
    for variant in ("plus_direct", "plus"):
        with pytest.raises(RuntimeError, match="invalid argument"):
            hcspmm.loi_reorder(_t(rp), _t(col), variant=variant)
    rp, col = np.load(os.path.join(GOLD, "loi_win_uniform_64.npz"))["row_pointers"], np.load(os.path.join(GOLD, "loi_win_uniform_64.npz"))["column_index"]
    hole = rp.copy()
    hole[6:] -= rp[6] - rp[5]  # row 5 loses its entries
    col_hole = np.concatenate([col[:rp[5]], col[rp[6]:]])
    with pytest.raises(RuntimeError, match="invalid argument"):
        hcspmm.loi_reorder(_t(hole), _t(col_hole), variant="plus_direct")
    r = int(np.argmax(np.diff(rp) >= 2))
    swapped = col.copy()
    swapped[rp[r]], swapped[rp[r] + 1] = col[rp[r] + 1], col[rp[r]]
    with pytest.raises(RuntimeError, match="invalid argument"):
        hcspmm.loi_reorder(_t(rp), _t(swapped), variant="plus")
    with pytest.raises(RuntimeError, match="invalid argument"):
        hcspmm.loi_reorder(_t(rp), _t(col), variant=7)


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "loi_*.npz")) if "loi_win_" not in p))
def test_loi_reorder_matches_reference_golden(path):
    g = np.load(path)
    perm, sizes = hcspmm.loi_reorder(_t(g["row_pointers"]), _t(g["column_index"]))
    assert np.array_equal(sizes.numpy(), g["group_sizes"])
    assert np.array_equal(perm.numpy(), g["order"])
    perm, sizes = hcspmm.loi_reorder(_t(g["row_pointers"]), _t(g["column_index"]), variant="new")
    assert np.array_equal(sizes.numpy(), g["group_sizes_new"]) and np.array_equal(perm.numpy(), g["order_new"])
    if "sym" in path:  # symmetric graphs: both variants coincide
        assert np.array_equal(g["order"], g["order_new"])


def test_loi_reorder_matches_compiled_reference_on_random_graphs():
    """Fuzz against the reference's own LOI.cpp, compiled where it lies into oracle/_ref/ (authoring container only:
    skipped wherever that binary does not exist, e.g. on the GPU box)."""
    ref_bin = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "loi_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/loi_ref not built (no /root/reference here)")
    sys.path.insert(0, GOLD)
    from make_loi_fixtures import run_ref
    rng = np.random.default_rng(77)
    for i in range(8):
        N = int(rng.integers(100, 4000))
        if i % 2 == 0:
            rp, col = graphs.powerlaw_graph(N, N * int(rng.integers(3, 25)), seed=100 + i,
                                            max_degree_frac=float(rng.choice([0.02, 0.3])))
            variants = ("new_direct", "new")
        else:
            rp, col = graphs.uniform_graph(N, N * int(rng.integers(1, 10)), seed=100 + i)
            variants = ("new_direct",)  # the other variant assumes a symmetric graph
        for variant in variants:
            sizes, _, order = run_ref(rp, col, variant)
            perm, gs = hcspmm.loi_reorder(_t(rp), _t(col), variant=variant)
            assert np.array_equal(perm.numpy(), order) and np.array_equal(gs.numpy(), sizes), (i, variant)
    # the windowed variants, on graphs inside the reference's defined domain (no row without entries)
    from make_loi_fixtures import _fill_empty_rows
    for i in range(12):
        N = int(rng.choice([50, 51, 64, 299, 300, 301, 650, 1500]))
        gen = (graphs.uniform_graph, graphs.powerlaw_graph)[i % 2]
        rp, col = _fill_empty_rows(*gen(N, N * int(rng.integers(2, 9)), seed=200 + i), seed=i)
        for variant in ("plus_direct", "plus"):
            sizes, _, order = run_ref(rp, col, variant)
            perm, gs = hcspmm.loi_reorder(_t(rp), _t(col), variant=variant)
            assert np.array_equal(perm.numpy(), order) and np.array_equal(gs.numpy(), sizes), (i, N, variant)


def test_loi_reorder_matches_oracle_on_larger_graph():
    from oracle import loi_oracle
    rp, col = graphs.powerlaw_graph(3000, 20000, seed=21)
    groups, visit = loi_oracle.reorder_new_direct(rp, col, len(rp) - 1)
    perm, sizes = hcspmm.loi_reorder(_t(rp), _t(col))
    assert np.array_equal(perm.numpy(), loi_oracle.final_order(groups, visit))
    assert sizes.tolist() == [len(x) for x in groups]


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "loi_*.npz")) if "loi_win_" not in p))
def test_fast_loi_with_both_relaxations_off_is_the_reference_permutation(path):
    """hcspmm_loi_reorder_fast is a different implementation (bitmap placement state, hashed candidate tables, staged growth)
    of the same greedy: one seed at a time and uncapped list walks must reproduce the fixtures the reference's LOI.cpp made."""
    g = np.load(path)
    perm, sizes = hcspmm.loi_reorder(_t(g["row_pointers"]), _t(g["column_index"]), variant="fast", batch=1, list_cap=-1)
    assert np.array_equal(perm.numpy(), g["order"]) and np.array_equal(sizes.numpy(), g["group_sizes"])


def test_fast_loi_is_a_permutation_independent_of_the_thread_count():
    for gen, args in ((graphs.community_graph, (60000, 125000)), (graphs.powerlaw_graph, (9000, 150000))):
        rp, col = gen(*args, seed=5)[:2]
        N = len(rp) - 1
        base = None
        for threads in (1, 2, 5):
            perm, sizes = hcspmm.loi_reorder(_t(rp), _t(col), variant="fast", threads=threads)
            assert np.array_equal(np.sort(perm.numpy()), np.arange(N))
            assert int(sizes.sum()) == int((np.diff(rp) > 0).sum()) and sizes.min() >= 1 and sizes.max() <= 16
            if base is None:
                base = perm
            assert torch.equal(base, perm), threads
        # an explicit small batch and cap: another permutation, just as deterministic
        a = hcspmm.loi_reorder(_t(rp), _t(col), variant="fast", batch=7, list_cap=5, threads=1)[0]
        b = hcspmm.loi_reorder(_t(rp), _t(col), variant="fast", batch=7, list_cap=5, threads=4)[0]
        assert torch.equal(a, b) and np.array_equal(np.sort(a.numpy()), np.arange(N))


def test_fast_loi_parameter_corners():
    """Tiny graphs, more seeds per round than vertices, caps of 1, one seed at a time: always a permutation that covers every non-empty
    row with a group, the same for any thread count, and the exact order when both relaxations are off."""
    rng = np.random.default_rng(0)
    for trial in range(16):
        N = int(rng.choice([1, 2, 17, 100, 999, 4096, 6000]))
        if trial % 4 == 0:
            rp, col = graphs.powerlaw_graph(max(N, 8), max(N, 8) * int(rng.integers(1, 12)), seed=trial, max_degree_frac=float(rng.choice([0.02, 0.9])))
        elif trial % 4 == 1:
            rp, col = graphs.uniform_graph(max(N, 4), max(N, 4) * int(rng.integers(0, 6)) + 1, seed=trial)
        elif trial % 4 == 2:
            rp, col = graphs.molecule_graph(max(N, 50), seed=trial)
        else:
            rp, col = graphs.community_graph(max(N, 64), max(N, 64) * 2, seed=trial)[:2]
        n = len(rp) - 1
        for kw in (dict(), dict(batch=int(rng.integers(1, 50)), list_cap=int(rng.choice([-1, 1, 3, 64]))), dict(batch=n + 5), dict(batch=1, list_cap=-1)):
            a, b = (hcspmm.loi_reorder(_t(rp), _t(col), variant="fast", threads=t, **kw) for t in (1, 3))
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (trial, kw)
            assert np.array_equal(np.sort(a[0].numpy()), np.arange(n)) and int(a[1].sum()) == int((np.diff(rp) > 0).sum()), (trial, kw)
            if kw == dict(batch=1, list_cap=-1):
                pe, se = hcspmm.loi_reorder(_t(rp), _t(col))
                assert torch.equal(pe, a[0]) and torch.equal(se, a[1]), trial


def test_fast_loi_recovers_planted_communities():
    """community_graph hides groups of 8-40 rows sharing a column pool behind shuffled vertex ids; after the relaxed reorder most
    16-row windows must again come from one or two planted groups, about as many as after the exact reorder."""
    rp, col, grp = graphs.community_graph(80000, 167000, seed=9)
    N = len(rp) - 1

    def pure_windows(perm):
        w = grp[np.asarray(perm)][: N // 16 * 16].reshape(-1, 16)
        return float(np.mean([len(np.unique(r)) <= 2 for r in w]))
    assert pure_windows(np.arange(N)) < 0.01
    fast = pure_windows(hcspmm.loi_reorder(_t(rp), _t(col), variant="fast")[0].numpy())
    exact = pure_windows(hcspmm.loi_reorder(_t(rp), _t(col))[0].numpy())
    assert fast > 0.6 and fast > exact - 0.05, (fast, exact)
    # and the classifier sees it: windows on the dense-tile path before / after
    before = int(_pre(rp, col)[3].sum())
    rpr, colr = hcspmm.apply_permutation(_t(rp), _t(col), hcspmm.loi_reorder(_t(rp), _t(col), variant="fast")[0])
    after = int(_pre(rpr.numpy(), colr.numpy())[3].sum())
    assert before < 0.02 * (N // 16) and after > 0.8 * (N // 16), (before, after)


def test_fast_loi_refuses_malformed_graphs(capi):
    rp, col = graphs.powerlaw_graph(5000, 30000, seed=2)
    bad = col.copy()
    bad[17] = 5000
    with pytest.raises(RuntimeError):
        hcspmm.loi_reorder(_t(rp), _t(bad), variant="fast")
    rp_bad = rp.copy()
    rp_bad[10] = rp_bad[11] + 1
    with pytest.raises(RuntimeError):
        hcspmm.loi_reorder(_t(rp_bad), _t(col), variant="fast")
    with pytest.raises(RuntimeError):
        hcspmm.loi_reorder(_t(rp), _t(col), variant="fast", batch=-1)
    perm, sizes = hcspmm.loi_reorder(_t(np.zeros(1, np.int32)), _t(np.zeros(0, np.int32)), variant="fast")
    assert perm.numel() == 0 and sizes.numel() == 0


def test_apply_permutation_parallel_path_matches_scipy():
    import scipy.sparse as sp
    rp, col = graphs.powerlaw_graph(30000, 400000, seed=8)  # enough entries for several host threads
    N = len(rp) - 1
    p = np.random.default_rng(1).permutation(N).astype(np.int32)
    rp2, col2 = hcspmm.apply_permutation(_t(rp), _t(col), _t(p))
    A = sp.csr_matrix((np.ones(len(col), np.int8), col, rp), shape=(N, N))
    B = A[p][:, p].tocsr()
    B.sort_indices()
    assert np.array_equal(rp2.numpy(), B.indptr) and np.array_equal(col2.numpy(), B.indices)


def test_loi_program_writes_the_reference_order_file(tmp_path, capsys):
    """hc-spmm_amd/LOI.py, the counterpart of LOI.cpp's main (LOI.cpp:807-896): dataset text file in, reorder_direct.txt out -- one old
    vertex id per line in the reference's order -- plus the relabelled graph on request."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hc_spmm_loi_program", os.path.join(ROOT, "hc-spmm_amd", "LOI.py"))
    prog = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(prog)
    g = np.load(os.path.join(GOLD, "loi_powerlaw_sym_600.npz"))
    rp, col = g["row_pointers"], g["column_index"]
    src = str(tmp_path / "graph.txt")
    graphs.write_coo_text(src, rp, col)
    out, moved = str(tmp_path / "reorder_direct.txt"), str(tmp_path / "graph_loi.txt")
    prog.main([src, "--out", out, "--apply", moved])
    printed = capsys.readouterr().out.split("\n")
    assert printed[0].startswith("All time: ") and int(printed[1]) == int((g["group_sizes"] == 16).sum())
    order = np.loadtxt(out, dtype=np.int64)
    assert np.array_equal(order, g["order"])  # (every vertex of this graph has an entry, so the text file carries all 600)
    rp2, col2 = prog.read_csr(moved)
    want_rp, want_col = hcspmm.apply_permutation(_t(rp), _t(col), _t(order.astype(np.int32)))
    # (the reorder puts the rows without entries last, and a text file of entries cannot name trailing empty rows: the reader sees the
    # graph up to its last non-empty row)
    k = len(rp2)
    assert np.array_equal(rp2, want_rp.numpy()[:k]) and bool((want_rp.numpy()[k - 1:] == len(col)).all()) and np.array_equal(col2, want_col.numpy())
    # the relaxed variant through the same program
    prog.main([src, "--out", out, "--variant", "fast"])
    assert sorted(np.loadtxt(out, dtype=np.int64).tolist()) == list(range(len(rp) - 1))


def test_apply_permutation_is_a_graph_isomorphism():
    import scipy.sparse as sp
    rp, col = graphs.powerlaw_graph(700, 5000, seed=6)
    N = len(rp) - 1
    perm, _ = hcspmm.loi_reorder(_t(rp), _t(col))
    rp2, col2 = hcspmm.apply_permutation(_t(rp), _t(col), perm)
    A = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(N, N))
    B = sp.csr_matrix((np.ones(len(col)), col2.numpy(), rp2.numpy()), shape=(N, N))
    p = perm.numpy()
    assert (A[p][:, p] != B).nnz == 0
    for r in range(N):
        assert np.all(np.diff(col2.numpy()[rp2[r]:rp2[r + 1]]) > 0)
    # the reorder must not lose dense-path windows on a graph that has planted structure
    rp3, col3 = graphs.planted_dense_graph(1600, seed=2)
    shuffle = np.random.default_rng(0).permutation(1600).astype(np.int32)
    rps, cols = hcspmm.apply_permutation(_t(rp3), _t(col3), _t(shuffle))
    before = int(_pre(rps.numpy(), cols.numpy())[3].sum())
    perm2, _ = hcspmm.loi_reorder(rps, cols)
    rpr, colr = hcspmm.apply_permutation(rps, cols, perm2)
    after = int(_pre(rpr.numpy(), colr.numpy())[3].sum())
    assert after > before


def test_forward_requires_gpu_tensors():
    rp, col = graphs.powerlaw_graph(64, 200, seed=1)
    bp, e2c, e2r, ht, rn, cn = _pre(rp, col)
    X = torch.zeros(64, 8)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        hcspmm.forward(X, _t(rp), _t(col), bp, e2c, e2r, ht, rn, cn)


@pytest.mark.parametrize("name", ["powerlaw_777", "planted_640", "uniform_300"])
def test_preprocess_frozen_fixture(oracle_mod, name):
    """Oracle AND product against the committed integer fixture (tests/golden/preprocess_ints.npz)."""
    g = np.load(os.path.join(GOLD, "preprocess_ints.npz"))
    rp, col = g[name + "_row_pointers"], g[name + "_column_index"]
    for rule in (0, 1, 2, 3, 4):
        o = oracle_mod.preprocess(rp, col, rule)
        p = _pre(rp, col, rule)
        for got in (o, [t.numpy() for t in p[:4]]):
            assert np.array_equal(got[0], g["%s_rule%d_blockPartition" % (name, rule)])
            assert np.array_equal(got[3], g["%s_rule%d_hybrid_type" % (name, rule)])
            assert np.array_equal(got[1], g[name + "_edgeToColumn"]) and np.array_equal(got[2], g[name + "_edgeToRow"])


def test_plan_column_slices_layout():
    """hcspmm_plan_params.slice_threshold / n_slices: rows longer than the threshold are cut at column-slice boundaries;
    every slice's pieces lie in its own column range, the ranges ascend with the slice number, a row's pieces take
    consecutive partial slots in CSR order, and every entry is covered exactly once."""
    rp, col = graphs.powerlaw_graph(3000, 90000, seed=5, max_degree_frac=0.3)
    N, E = len(rp) - 1, len(col)
    bp, e2c, e2r, ht, _, _ = _pre(rp, col, 2)
    for S, thr, seg in ((8, 16, 0), (16, 40, 32), (8, 1, 0)):
        plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, slice_threshold=thr, n_slices=S,
                                 segment_len=seg, split_threshold=max(seg, 0) * 2).numpy()
        h, tasks, _, fix = _decode_plan(plan)
        assert h.n_slices == S and h.slice_threshold == thr and h.total_words == len(plan)
        assert hcspmm.capi.lib().hcspmm_plan_check(ctypes.byref(h), N, E, len(plan)) == 0
        table, slices = _decode_slices(plan, h)
        deg = np.diff(rp)
        assert h.n_sliced_rows == int((deg > thr).sum()) and h.nnz_sliced == int(deg[deg > thr].sum())
        cover = np.zeros(E, np.int32)
        free = _expand_tiny(h, tasks, fix, rp, col)
        for row, e0, ln, slot in free:
            assert deg[row] <= thr
            cover[e0:e0 + ln] += 1
        lo_hi = []
        pieces = {}
        for s, d in enumerate(slices):
            if len(d):
                lo_hi.append((int(col[d[:, 1]].min()), int(col[d[:, 1] + d[:, 2] - 1].max())))
            for row, e0, ln, slot in d:
                assert deg[row] > thr and rp[row] <= e0 and e0 + ln <= rp[row + 1]
                cover[e0:e0 + ln] += 1
                pieces.setdefault(int(row), []).append((int(e0), int(ln), int(slot)))
        assert np.all(cover == 1)
        assert all(a[1] < b[0] for a, b in zip(lo_hi, lo_hi[1:]))  # the slices' column ranges ascend and do not overlap
        fixmap = {int(r): (int(s0), int(ns)) for r, s0, ns, _ in fix}
        n_slots = 0
        for row, ps in pieces.items():
            ps.sort()
            if len(ps) == 1:
                assert ps[0][2] == -1 and row not in fixmap
                continue
            s0, ns = fixmap[row]
            assert ns == len(ps) and [p[2] for p in ps] == list(range(s0, s0 + ns))  # slot order = CSR order
            n_slots += ns
        assert h.n_partials == n_slots and h.n_split_rows == len(fixmap)
    # off: no slice sections, the round-2 layout
    plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, slice_threshold=-1).numpy()
    h = _decode_plan(plan)[0]
    assert h.n_slices == 0 and h.n_slice_tasks == 0 and h.total_words == len(plan)


def test_plan_column_slices_automatic_rule():
    """Automatic mode: on only when X spans many L2s (num_columns >= 65536; below 250 000 columns only for >= 3 M sparse-path
    entries) and the long rows hold >= 5 % of the entries."""
    rp, col = graphs.powerlaw_graph(3000, 90000, seed=5, max_degree_frac=0.3)
    bp, e2c, e2r, ht, plan, _ = _pre(rp, col, 2)
    assert _decode_plan(plan.numpy())[0].n_slices == 0  # 3000 columns
    plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, num_columns=70000).numpy()
    assert _decode_plan(plan)[0].n_slices == 0  # 70 000 columns but only 90 K entries: a launch this small loses to the extra region
    plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, num_columns=250000).numpy()
    h = _decode_plan(plan)[0]
    assert h.n_slices == 8 and h.slice_threshold == 256 and h.total_words == len(plan)
    rp2, col2 = graphs.uniform_graph(3000, 30000, seed=1)  # no long rows
    bp, e2c, e2r, ht, _, _ = _pre(rp2, col2, 2)
    assert _decode_plan(hcspmm.build_plan(torch.from_numpy(rp2), torch.from_numpy(col2), bp, e2c, ht, num_columns=250000).numpy())[0].n_slices == 0


def test_plan_panel_cols_parameter_lands_in_the_header():
    rp, col = graphs.powerlaw_graph(500, 3000, seed=1)
    bp, e2c, e2r, ht, _, _ = _pre(rp, col)
    for asked, stored in ((0, 0), (32, 32), (40, 48), (64, 64), (-3, -1), (-1, -1)):
        plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, panel_cols=asked).numpy()
        h = _decode_plan(plan)[0]
        assert h.panel_cols == stored and hcspmm.capi.lib().hcspmm_plan_check(ctypes.byref(h), len(rp) - 1, len(col), len(plan)) == 0
