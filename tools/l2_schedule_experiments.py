"""Two schedules for the rows the column slices leave alone, tried in the per-XCD L2 LRU model (tools/l2_hit_simulation.py,
Reddit-scale headline) before spending GPU time -- neither is built:
  levels : nested slice levels -- 8 slices for the longest rows, 4 / 2 "super-slices" (served by XCD pairs / quads) for
           mid-length rows, so that those pay 4 / 2 partial rows instead of 8;
  affine : rows of at most the slice threshold entries go WHOLE (no partial sums) to the XCD that owns the slice holding
           most of their entries.
  python tools/l2_schedule_experiments.py            (log: profiles/r03/l2_schedule_experiments.log)"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "hc-spmm_amd"), os.path.join(R, "tools")]
import numpy as np  # noqa: E402
import l2_hit_simulation as L  # noqa: E402
from hcspmm import graphs  # noqa: E402

rp, col = graphs.powerlaw_graph(233000, 11600000, seed=3)
N, E = len(rp) - 1, len(col)
deg = np.diff(rp).astype(np.int64)
rows = np.arange(N, dtype=np.int64)
lib = L._lib()
TPW = 32  # tasks per workgroup (4 waves x 8 lane groups at a 32-column panel)


def slice_bounds(min_len):
    """8-way column boundaries with equal shares of the entries of rows longer than min_len; entries of those rows."""
    sl_rows = rows[deg > min_len]
    ent_row = np.repeat(sl_rows, deg[sl_rows])
    ent = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in sl_rows])
    c = col[ent].astype(np.int64)
    hist = np.bincount(c, minlength=N).cumsum()
    return np.searchsorted(hist, hist[-1] * np.arange(1, 8) / 8), ent_row, ent, c


def pieces(er, ee, cc, bounds, S, seg=256):
    sl = np.searchsorted(bounds, cc, side="right")
    key = er * S + sl
    start = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    p_len = np.diff(np.concatenate([start, [len(key)]]))
    q_row, q_e0, q_len = L.cut_segments(er[start], ee[start], p_len, seg)
    return q_row, q_e0, q_len, np.repeat(sl[start], (p_len + seg - 1) // seg)


def deal(e0, ln, xcds, per_xcd, kind):
    """class-ordered tasks dealt workgroup by workgroup over the XCDs `xcds`"""
    nwg = (len(e0) + TPW - 1) // TPW
    for i, x in enumerate(xcds):
        wgs = np.arange(i, nwg, len(xcds))
        idx = (wgs[:, None] * TPW + np.arange(TPW)[None, :]).ravel()
        idx = idx[idx < len(e0)]
        per_xcd[x].append((e0[idx], ln[idx], np.full(len(idx), kind, np.int8)))


def finish(per_xcd, free_rows):
    o = L.class_order(free_rows, deg[free_rows])
    f_e0, f_len = rp[free_rows][o].astype(np.int64), deg[free_rows][o]
    deal(f_e0, f_len, list(range(8)), per_xcd, 0)
    out = []
    for x in range(8):
        e0 = np.concatenate([p[0] for p in per_xcd[x]]).astype(np.int32)
        ln = np.concatenate([p[1] for p in per_xcd[x]]).astype(np.int32)
        kd = np.concatenate([p[2] for p in per_xcd[x]])
        ws = np.minimum(np.arange(0, len(e0) + TPW, TPW), len(e0)).astype(np.int64)
        out.append((e0, ln, kd, ws if ws[-1] == len(e0) else np.append(ws, len(e0))))
    return out


def levels(spec):
    """spec: [(min_len_exclusive, S)] by min_len descending."""
    b8, ent_row, ent, c = slice_bounds(min(l[0] for l in spec))
    per_xcd, n_part, assigned = [[] for _ in range(8)], 0, np.zeros(N, bool)
    for mn, S in spec:
        m_rows = (deg > mn) & ~assigned
        assigned |= m_rows
        em = m_rows[ent_row]
        q_row, q_e0, q_len, q_sl = pieces(ent_row[em], ent[em], c[em], b8[(8 // S) - 1::(8 // S)] if S < 8 else b8, S)
        n_part += int((np.bincount(q_row, minlength=N)[q_row] > 1).sum())
        for s in range(S):
            m = q_sl == s
            o = L.class_order(q_row[m], q_len[m])
            deal(q_e0[m][o], q_len[m][o], list(range(s * 8 // S, (s + 1) * 8 // S)), per_xcd, 1)
    return finish(per_xcd, rows[~assigned & (deg > 0)]), n_part


def affine(thr, affine_min):
    b8, ent_row, ent, c = slice_bounds(thr)
    q_row, q_e0, q_len, q_sl = pieces(ent_row, ent, c, b8, 8)
    per_xcd = [[] for _ in range(8)]
    for s in range(8):
        m = q_sl == s
        o = L.class_order(q_row[m], q_len[m])
        per_xcd[s].append((q_e0[m][o], q_len[m][o], np.ones(m.sum(), np.int8)))
    cnt = np.zeros((N, 8), np.int32)
    np.add.at(cnt, (np.repeat(rows, deg), np.searchsorted(b8, np.arange(N), side="right")[col]), 1)
    aff = rows[(deg <= thr) & (deg > affine_min)]
    dom = np.argmax(cnt[aff], axis=1)
    for s in range(8):
        r = aff[dom == s]
        o = L.class_order(r, deg[r])
        per_xcd[s].append((rp[r][o].astype(np.int64), deg[r][o], np.zeros(len(r), np.int8)))
    share = cnt[aff, dom].sum() / max(deg[aff].sum(), 1)
    return finish(per_xcd, rows[(deg <= min(thr, affine_min)) & (deg > 0)]), share


def report(tag, sched, extra=""):
    hits, tot = L.simulate(lib, col, N, sched)
    x = np.array([int(s[1].sum()) for s in sched], float)
    print("%-44s L2 hit all %.3f  sliced %.3f  others %.3f  busiest XCD / mean %.3f  %s" % (
        tag, hits.sum() / tot.sum(), hits[1] / max(tot[1], 1), hits[0] / max(tot[0], 1), x.max() / x.mean(), extra), flush=True)


for spec in ([(256, 8)], [(128, 8)], [(256, 8), (64, 4)], [(256, 8), (64, 4), (16, 2)], [(256, 8), (64, 2)], [(256, 8), (128, 4), (32, 2)]):
    sched, n_part = levels(spec)
    miss_note = "partial rows %d" % n_part
    report("levels %s" % spec, sched, miss_note)
for thr, amin in ((256, 10 ** 9), (256, 2), (256, 32), (128, 2)):
    sched, share = affine(thr, amin)
    report("slices > %d; whole rows > %s by dominant slice" % (thr, "-" if amin > 10 ** 6 else amin), sched,
           "share of their entries in the dominant slice %.3f" % share)
