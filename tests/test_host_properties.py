"""Property tests (hypothesis) of the host side of the path on small ragged graphs -- empty rows, N < 16, N % 16 != 0,
hub rows, duplicate-free random rows: preprocess integers == oracle for every rule; the plan blob accounts for every
stored entry exactly once (sparse tasks + dense windows), lists exactly the non-dense windows, carries the graph's
fingerprint, and passes its own consistency check; LOI reorders are permutations that apply as graph isomorphisms."""
import ctypes

import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import hcspmm
from hcspmm.capi import Header

from test_host_cpu import _decode_plan, _decode_slices, _expand_tiny


@st.composite
def csr_graphs(draw):
    N = draw(st.integers(1, 90))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    shape = draw(st.sampled_from(["sparse", "dense_windows", "hubs", "mostly_empty"]))
    rng = np.random.default_rng(seed)
    if shape == "sparse":
        deg = rng.integers(0, min(N, 6) + 1, N)
    elif shape == "mostly_empty":
        deg = (rng.random(N) < 0.2) * rng.integers(1, min(N, 4) + 1, N)
    elif shape == "hubs":
        deg = rng.integers(0, 3, N)
        deg[rng.integers(0, N, 2)] = N
    else:
        deg = rng.integers(0, min(N, 10) + 1, N)
    deg = np.minimum(deg, N)
    cols = []
    for w0 in range(0, N, 16):
        pool = rng.choice(N, min(N, int(rng.integers(1, 25))), replace=False) if shape == "dense_windows" else None
        for r in range(w0, min(w0 + 16, N)):
            d = int(deg[r])
            if pool is not None:
                d = min(d, len(pool))
                deg[r] = d
                cols.append(np.sort(rng.choice(pool, d, replace=False)))
            else:
                cols.append(np.sort(rng.choice(N, d, replace=False)))
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = (np.concatenate(cols) if cols else np.zeros(0)).astype(np.int32)
    return rp, col


# derandomize: the same examples on every run (a CI tier must not depend on the draw); widen locally with HYPOTHESIS_SEED / max_examples
SETTINGS = dict(max_examples=60, deadline=None, derandomize=True, database=None,
                suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@settings(**SETTINGS)
@given(csr_graphs(), st.sampled_from([0, 1, 2, 3, 4]))
def test_preprocess_equals_oracle(oracle_mod, g, rule):
    rp, col = g
    N = len(rp) - 1
    got = hcspmm.preprocess(torch.from_numpy(col), torch.from_numpy(rp), N, len(col), (N + 15) // 16, rule=rule)
    want = oracle_mod.preprocess(rp, col, rule)
    for w, t in zip(want, got[:4]):
        assert np.array_equal(w, t.numpy())


@settings(**SETTINGS)
@given(csr_graphs(), st.sampled_from([0, 2, "all_dense"]), st.integers(2, 6), st.integers(1, 6))
def test_plan_covers_every_entry_exactly_once(capi, g, mode, split, seg):
    rp, col = g
    N, E = len(rp) - 1, len(col)
    W = (N + 15) // 16
    bp, e2c, e2r, ht, _, _ = hcspmm.preprocess(torch.from_numpy(col), torch.from_numpy(rp), N, E, W, rule=0 if mode == "all_dense" else mode)
    if mode == "all_dense":
        ht = torch.ones_like(ht)
    plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, split_threshold=split, segment_len=min(seg, split)).numpy()
    h, tasks, dindex, fix = _decode_plan(plan)
    L = capi.lib()
    assert L.hcspmm_plan_check(ctypes.byref(h), N, E, len(plan)) == 0
    fp = ctypes.c_uint64(0)
    assert L.hcspmm_graph_fingerprint_host(rp.ctypes.data, col.ctypes.data, N, E, ctypes.byref(fp)) == 0 and fp.value == h.fingerprint
    cover = np.zeros(E, np.int32)
    for row, e0, ln, slot in _expand_tiny(h, tasks, fix, rp, col):
        assert rp[row] <= e0 and e0 + ln <= rp[row + 1]
        cover[e0:e0 + ln] += 1
    win_nnz = np.array([rp[min(16 * w + 16, N)] - rp[16 * w] for w in range(W)])
    dense_w = [w for w in range(W) if ht[w] != 0 and win_nnz[w] > 0]
    assert sorted(dindex[:, 0].tolist()) == dense_w and h.n_dense == len(dense_w)
    for w in dense_w:
        cover[rp[16 * w]:rp[min(16 * w + 16, N)]] += 1
    assert np.all(cover == 1)
    sparse_list = plan[h.off_sparse_windows:h.off_sparse_windows + h.n_sparse_windows].tolist()
    assert sparse_list == [w for w in range(W) if w not in set(dense_w)]
    assert h.nnz_dense == int(win_nnz[dense_w].sum()) and h.nnz_sparse == E - h.nnz_dense
    assert h.dense_k_sum == int(sum(8 * int(bp[w]) for w in dense_w))
    # split rows: every row longer than the threshold on the sparse path has a fix-up entry covering its segments
    deg = np.diff(rp)
    long_rows = [r for r in range(N) if deg[r] > split and (r // 16) not in set(dense_w)]
    assert sorted(fix[:, 0].tolist()) == long_rows
    for row, s0, ns, _ in fix:
        assert ns == -(-deg[row] // min(seg, split))


@settings(**SETTINGS)
@given(csr_graphs(), st.sampled_from([0, 2]), st.integers(1, 9), st.sampled_from([8, 16, 24, 64]), st.sampled_from([0, 1, 3, 7]))
def test_sliced_plan_covers_every_entry_exactly_once(capi, g, rule, thr, S, seg):
    """XCD-affine column slices forced on ragged graphs: free tasks + slice pieces + dense windows cover every stored entry
    exactly once; a slice's pieces lie in its own ascending column range; a row's pieces take consecutive partial slots in
    CSR order; the header's counts are the real ones (the blob is sized by upper bounds)."""
    rp, col = g
    N, E = len(rp) - 1, len(col)
    W = (N + 15) // 16
    bp, e2c, e2r, ht, _, _ = hcspmm.preprocess(torch.from_numpy(col), torch.from_numpy(rp), N, E, W, rule=rule)
    plan = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, slice_threshold=thr, n_slices=S,
                             segment_len=seg, split_threshold=2 * seg).numpy()
    h, tasks, dindex, fix = _decode_plan(plan)
    assert capi.lib().hcspmm_plan_check(ctypes.byref(h), N, E, len(plan)) == 0 and h.total_words == len(plan)
    deg = np.diff(rp)
    win_nnz = np.array([rp[min(16 * w + 16, N)] - rp[16 * w] for w in range(W)])
    dense_rows = np.repeat((ht.numpy() != 0) & (win_nnz > 0), 16)[:N]
    n_long = int(((deg > thr) & ~dense_rows).sum())
    assert h.n_slices == (S if n_long else 0) and h.n_sliced_rows == n_long
    table, slices = _decode_slices(plan, h)
    cover = np.zeros(E, np.int32)
    for row, e0, ln, slot in _expand_tiny(h, tasks, fix, rp, col):
        assert deg[row] <= thr or h.n_slices == 0
        cover[e0:e0 + ln] += 1
    ranges, pieces = [], {}
    for d in slices:
        if len(d):
            ranges.append((int(col[d[:, 1]].min()), int(col[d[:, 1] + d[:, 2] - 1].max())))
        for row, e0, ln, slot in d:
            assert deg[row] > thr and rp[row] <= e0 and e0 + ln <= rp[row + 1]
            cover[e0:e0 + ln] += 1
            pieces.setdefault(int(row), []).append((int(e0), int(ln), int(slot)))
    for w in dindex[:, 0]:
        cover[rp[16 * w]:rp[min(16 * w + 16, N)]] += 1
    assert np.all(cover == 1)
    assert all(a[1] < b[0] for a, b in zip(ranges, ranges[1:]))
    fixmap = {int(r): (int(s0), int(ns)) for r, s0, ns, _ in fix}
    for row, ps in pieces.items():
        ps.sort()
        if len(ps) == 1:
            assert ps[0][2] == -1 and row not in fixmap
        else:
            s0, ns = fixmap[row]
            assert ns == len(ps) and [q[2] for q in ps] == list(range(s0, s0 + ns))
    assert h.nnz_sliced == int(deg[(deg > thr) & ~dense_rows].sum()) if h.n_slices else h.nnz_sliced == 0


@settings(**SETTINGS)
@given(csr_graphs(), st.sampled_from([0, 2]), st.sampled_from([0, 3, 9]), st.sampled_from([2, 5, 40]), st.sampled_from([32, 64, 128]))
def test_row_tile_form_writes_every_out_row_exactly_once(capi, g, rule, thr, split, D):
    """The row-tile form of the fused operators (csrc/fused_rows.hip, capi.hip hcspmm_forward_fused) writes `out` from three
    places, each enumerating rows from the plan: the tile launches (whole rows among the ordinary tasks behind the n_wide
    longest, and among the tiny tasks; every row of a dense window), and dense_update_rows_kernel (whole rows among the n_wide
    longest tasks, every fix-up entry, slice descriptors with a direct store).  Whatever the split / slice / wide
    thresholds: those sets are disjoint and their union is [0, N)."""
    rp, col = g
    N, E = len(rp) - 1, len(col)
    W = (N + 15) // 16
    bp, e2c, e2r, ht, _, _ = hcspmm.preprocess(torch.from_numpy(col), torch.from_numpy(rp), N, E, W, rule=rule)
    plan_t = hcspmm.build_plan(torch.from_numpy(rp), torch.from_numpy(col), bp, e2c, ht, slice_threshold=thr if thr else -1,
                               split_threshold=split, segment_len=max(1, split // 2), fuse_in_launch=2, panel_cols=-1)
    plan = plan_t.numpy()
    h, tasks, dindex, fix = _decode_plan(plan)
    assert h.flags == 2
    thr_wide = capi.lib().hcspmm_wide_threshold(ctypes.byref(h), D)
    first_tiny = h.n_tasks - h.n_tiny
    n_wide = int((tasks[:first_tiny, 2] > thr_wide).sum()) if first_tiny else 0
    assert np.all(tasks[:n_wide, 2] > thr_wide)  # the wide tasks are a prefix of the (length-sorted) list
    written = np.zeros(N, np.int32)
    for row, e0, ln, slot in tasks[n_wide:first_tiny]:      # tile launch, ordinary tasks
        if slot < 0:
            written[row] += 1
    for x, i0, ln, i1 in tasks[first_tiny:]:                # tile launch, tiny tasks
        if x >= 0:
            written[x] += 1
    for w in dindex[:, 0]:                                   # tile launch, dense windows
        written[16 * w:min(16 * w + 16, N)] += 1
    for row, e0, ln, slot in tasks[:n_wide]:                 # leftover launch: whole rows among the wide tasks
        if slot < 0:
            written[row] += 1
    for row in fix[:, 0]:                                    # ... rows completed by the fix-up pass
        written[row] += 1
    if h.n_slices:                                           # ... sliced rows that fell into one piece (direct store)
        desc = plan[h.off_slice_tasks:h.off_slice_tasks + 4 * h.n_slice_tasks].reshape(-1, 4)
        for row, e0, ln, slot in desc:
            if row >= 0 and slot < 0:
                written[row] += 1
    assert np.all(written == 1), np.nonzero(written != 1)[0][:10]


@settings(**SETTINGS)
@given(csr_graphs(), st.sampled_from(["new_direct", "new"]))
def test_loi_reorder_is_a_permutation_and_relabelling_is_an_isomorphism(g, variant):
    rp, col = g
    N = len(rp) - 1
    perm, sizes = hcspmm.loi_reorder(torch.from_numpy(rp), torch.from_numpy(col), variant=variant)
    p = perm.numpy()
    assert sorted(p.tolist()) == list(range(N))
    # every row with entries is placed in a group of 1..16 rows (the `new` variant, which walks the column vertex's own
    # out-list, may also pull rows WITHOUT entries into groups of an asymmetric graph); the rest follows ascending
    n_grouped = int(sizes.sum())
    assert int((np.diff(rp) > 0).sum()) <= n_grouped <= N and (sizes.numpy() <= 16).all() and (sizes.numpy() >= 1).all()
    if variant == "new_direct":
        assert n_grouped == int((np.diff(rp) > 0).sum())
    tail = p[n_grouped:]
    assert np.all(np.diff(tail) > 0) and np.all(np.diff(rp)[tail] == 0)
    rp2, col2 = hcspmm.apply_permutation(torch.from_numpy(rp), torch.from_numpy(col), perm)
    rp2, col2 = rp2.numpy(), col2.numpy()
    inv = np.empty(N, np.int64)
    inv[p] = np.arange(N)
    for new_r in range(N):
        old = p[new_r]
        assert sorted(inv[col[rp[old]:rp[old + 1]]].tolist()) == col2[rp2[new_r]:rp2[new_r + 1]].tolist()
