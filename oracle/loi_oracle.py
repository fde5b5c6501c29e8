"""loi_oracle -- TEST INFRASTRUCTURE ONLY.

Pure-Python restatement (small graphs only) of the reference's layout reorder
("LOI"/"LOA") as actually wired in its main: `reorder_plus_new_direct`
(/root/reference/LOI.cpp:660-805) and the output order of main (LOI.cpp:873-891).

Pinned: tests/golden/loi_*.npz hold permutations produced by the reference's own
LOI.cpp compiled in the authoring container (oracle/Makefile target `ref`,
driver oracle/ref_loi_driver.cpp, generator tests/golden/make_loi_fixtures.py);
tests/test_oracle_cpu.py checks this restatement against them bit for bit.
"""
import numpy as np


def build_in_csr(rowptr, col, N):
    """In-neighbour CSR by a row-major scan, so each list is ascending (LOI.cpp:826-841)."""
    rowptr_in = np.zeros(N + 1, np.int64)
    for c in col:
        rowptr_in[c + 1] += 1
    rowptr_in = np.cumsum(rowptr_in)
    fill = rowptr_in.copy()
    col_in = np.zeros(len(col), np.int64)
    for r in range(N):
        for e in range(rowptr[r], rowptr[r + 1]):
            c = col[e]
            col_in[fill[c]] = r
            fill[c] += 1
    return rowptr_in, col_in


def _residual(cols_sorted, out_nbrs):
    """Columns of `out_nbrs` (CSR order) not yet in the group's sorted column list
    (cal_resi_elements, LOI.cpp:60-73); returns (new sorted list, residual)."""
    have = set(cols_sorted)
    resi = [c for c in out_nbrs if c not in have]
    return sorted(cols_sorted + resi), resi


def reorder_new(rowptr, col, N):
    """reorder_plus_new (LOI.cpp:505-658): identical to the _direct variant except that the rows
    "referencing" a column c are taken from c's OWN out-list (LOI.cpp:556-557, :606-607) -- correct
    only for symmetric graphs, where both variants coincide."""
    return reorder_new_direct(rowptr, col, N, symmetric_shortcut=True)


def reorder_new_direct(rowptr, col, N, symmetric_shortcut=False):
    """-> (groups: list[list[int]], visit: list[bool]); LOI.cpp:660-805."""
    rowptr = [int(x) for x in rowptr]
    col = [int(x) for x in col]
    rowptr_in, col_in = (rowptr, col) if symmetric_shortcut else build_in_csr(rowptr, col, N)
    deg = [rowptr[i + 1] - rowptr[i] for i in range(N)]
    front = [i for i in range(N) if deg[i] > 0]                       # :665-670
    visit = [False] * N
    cns = [0] * N
    groups = []
    cur = 0
    f32 = np.float32

    def pick(cand, base_ones, base_rows, first):
        best, best_p = -1, f32(0.0)
        for v in cand:                                                  # insertion order (:723, :772)
            if visit[v]:
                continue
            ones = base_ones + deg[v]
            rows = (ones - cns[v]) if first else (base_rows + deg[v] - cns[v])   # :727 / :776
            p = f32(ones) / f32(rows)                                   # (float)ones / rows
            if p > best_p:                                              # strict: first-seen wins ties
                best, best_p = v, p
        return best

    while True:
        while cur < len(front) and visit[front[cur]]:                   # :699-706
            cur += 1
        if cur >= len(front):
            break
        seed = front[cur]
        grp = [seed]
        visit[seed] = True
        seen, cand = set(), []
        for e in range(rowptr[seed], rowptr[seed + 1]):                 # :710-720
            c = col[e]
            for j in range(rowptr_in[c], rowptr_in[c + 1]):
                r = int(col_in[j])
                if not visit[r]:
                    cns[r] += 1
                    if r not in seen:
                        seen.add(r)
                        cand.append(r)
        v = pick(cand, deg[seed], 0, True)
        if v == -1:                                                     # :737-740
            groups.append(grp)
            continue
        grp.append(v)
        visit[v] = True
        cols = list(col[rowptr[seed]:rowptr[seed + 1]])                 # :745-747
        cols, resi = _residual(cols, col[rowptr[v]:rowptr[v + 1]])      # :749
        ones, nrows = deg[seed] + deg[v], len(cols)                     # :751-752
        for _ in range(14):                                             # :754
            for c in resi:                                              # :759-769
                for j in range(rowptr_in[c], rowptr_in[c + 1]):
                    r = int(col_in[j])
                    if not visit[r]:
                        cns[r] += 1
                        if r not in seen:
                            seen.add(r)
                            cand.append(r)
            v = pick(cand, ones, nrows, False)
            if v == -1:
                break
            grp.append(v)
            visit[v] = True
            cols, resi = _residual(cols, col[rowptr[v]:rowptr[v + 1]])  # :791
            ones += deg[v]
            nrows = len(cols)
        for r in cand:                                                  # :797-801
            cns[r] = 0
        groups.append(grp)
    return groups, visit


def final_order(groups, visit):
    """Order written by the reference's main (LOI.cpp:873-891): full groups, then short groups,
    then never-visited vertices ascending."""
    order = []
    for g in groups:
        if len(g) == 16:
            order.extend(g)
    for g in groups:
        if len(g) < 16:
            order.extend(g)
    order.extend(i for i, v in enumerate(visit) if not v)
    return np.asarray(order, np.int32)
