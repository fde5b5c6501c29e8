"""LRU model of the eight per-XCD L2s on a gather workload (one 32-column panel = one 128-byte line per X row):
hit rate and estimated fabric traffic of the sparse-row path under (a) the round-2 task order and (b) XCD-affine
column slices -- rows longer than a threshold cut at S column boundaries, slice s served only by workgroups with
blockIdx % 8 == s % 8, so each L2 holds 1/8 (S = 8) of the X rows those tasks touch.  Host-only (numpy + a small C
helper compiled on the fly); results quoted in DESIGN.md.

  python tools/l2_hit_simulation.py [--workload reddit] [--thresholds 512,128,64] [--slices 8,16]
"""
import argparse
import ctypes
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "hc-spmm_amd")]
import numpy as np  # noqa: E402
from hcspmm import graphs  # noqa: E402


def _lib():
    src = os.path.join(R, "tools", "l2_lru_sim.c")
    out = os.path.join("/tmp", "l2_lru_sim_%d.so" % os.getuid())
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", out])
    lib = ctypes.CDLL(out)
    P = ctypes.c_void_p
    lib.simulate_xcd.argtypes = [P, ctypes.c_int64, P, P, P, P, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]
    return lib


def length_class(l):
    c = np.zeros(l.shape, np.int64)
    nz = l > 0
    c[nz] = np.ceil(np.log2(np.maximum(l[nz], 1))).astype(np.int64) + 1
    return c


def cut_segments(row, e0, ln, seg):
    """(row, e0, len) pieces -> pieces of at most seg entries."""
    n = (ln + seg - 1) // seg
    idx = np.repeat(np.arange(len(ln)), n)
    k = np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n)
    return row[idx], e0[idx] + k * seg, np.minimum(seg, ln[idx] - k * seg)


def class_order(row, ln):
    return np.lexsort((row, -length_class(ln)))


def build_schedule(rp, col, n_cols, thr, S, seg=256, split=512, tasks_per_wg=32, n_xcd=8, interleave=False):
    """-> per-XCD (e0, len, kind, wg_start) + statistics.  thr <= 0: the round-2 schedule (rows > split cut every seg entries)."""
    N = len(rp) - 1
    deg = np.diff(rp).astype(np.int64)
    rows = np.arange(N, dtype=np.int64)
    per_xcd = [dict(e0=[], ln=[], kind=[]) for _ in range(n_xcd)]
    stats = {}
    if thr > 0:
        sl_rows = rows[deg > thr]
        free_rows = rows[(deg <= thr) & (deg > 0)]
        # column boundaries: equal shares of the sliced rows' entries
        ent_row = np.repeat(sl_rows, deg[sl_rows])
        ent = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in sl_rows]) if len(sl_rows) else np.empty(0, np.int64)
        c = col[ent].astype(np.int64)
        hist = np.bincount(c, minlength=n_cols).cumsum()
        bounds = np.searchsorted(hist, hist[-1] * np.arange(1, S) / S)  # column id boundaries
        sl = np.searchsorted(bounds, c, side="right")
        # pieces = runs of equal (row, slice)
        key = ent_row * S + sl
        start = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
        p_len = np.diff(np.concatenate([start, [len(key)]]))
        p_row, p_e0, p_sl = ent_row[start], ent[start], sl[start]
        q_row, q_e0, q_len = cut_segments(p_row, p_e0, p_len, seg)
        q_sl = np.repeat(p_sl, (p_len + seg - 1) // seg)
        stats["sliced_rows"] = int(len(sl_rows))
        stats["sliced_nnz"] = int(len(ent))
        stats["n_partials"] = int(len(q_row))
        stats["slice_cols"] = np.diff(np.concatenate([[0], bounds, [n_cols]])).tolist()
        for s in range(S):
            m = q_sl == s
            o = class_order(q_row[m], q_len[m])
            x = s % n_xcd
            per_xcd[x]["e0"].append(q_e0[m][o])
            per_xcd[x]["ln"].append(q_len[m][o])
            per_xcd[x]["kind"].append(np.ones(m.sum(), np.int8))
        f_row, f_e0, f_len = free_rows, rp[free_rows].astype(np.int64), deg[free_rows]
    else:
        big = deg > split
        b_row, b_e0, b_len = cut_segments(rows[big], rp[:-1][big].astype(np.int64), deg[big], seg)
        small = (~big) & (deg > 0)
        f_row = np.concatenate([b_row, rows[small]])
        f_e0 = np.concatenate([b_e0, rp[:-1][small].astype(np.int64)])
        f_len = np.concatenate([b_len, deg[small]])
        stats["n_partials"] = int(len(b_row))
    # sliced regions of every XCD are padded to whole workgroups; then the free tasks, dealt workgroup by workgroup
    wg_lists = []
    for x in range(n_xcd):
        e0 = np.concatenate(per_xcd[x]["e0"]) if per_xcd[x]["e0"] else np.empty(0, np.int64)
        ln = np.concatenate(per_xcd[x]["ln"]) if per_xcd[x]["ln"] else np.empty(0, np.int64)
        kd = np.concatenate(per_xcd[x]["kind"]) if per_xcd[x]["kind"] else np.empty(0, np.int8)
        wg_lists.append([e0, ln, kd])
    o = class_order(f_row, f_len)
    f_e0, f_len = f_e0[o], f_len[o]
    n_free_wg = (len(f_e0) + tasks_per_wg - 1) // tasks_per_wg
    out = []
    for x in range(n_xcd):
        e0, ln, kd = wg_lists[x]
        n_sl_wg = (len(e0) + tasks_per_wg - 1) // tasks_per_wg
        starts = list(np.minimum(np.arange(n_sl_wg + 1) * tasks_per_wg, len(e0)))
        wgs = np.arange(x, n_free_wg, n_xcd)
        idx = (wgs[:, None] * tasks_per_wg + np.arange(tasks_per_wg)[None, :]).ravel()
        idx = idx[idx < len(f_e0)]
        fe0, fln = f_e0[idx], f_len[idx]
        base = len(e0)
        fst = base + np.minimum(np.arange(1, len(wgs) + 1) * tasks_per_wg, len(fe0))
        e0 = np.concatenate([e0, fe0]).astype(np.int32)
        ln = np.concatenate([ln, fln]).astype(np.int32)
        kd = np.concatenate([kd, np.zeros(len(fe0), np.int8)])
        wg_start = np.array(starts + list(fst), dtype=np.int64)
        if interleave and n_sl_wg and len(wgs):
            # sliced and free workgroups alternate in proportion (both resources -- L2 hits and fabric misses -- busy at once)
            n_a, n_b = n_sl_wg, len(wgs)
            pos_a = (np.arange(n_a) + 0.5) / n_a
            pos_b = (np.arange(n_b) + 0.5) / n_b
            order = np.argsort(np.concatenate([pos_a, pos_b]), kind="stable")
            sizes = np.diff(wg_start)
            new_e0, new_ln, new_kd, new_start = [], [], [], [0]
            for w in order:
                a, b = wg_start[w], wg_start[w + 1]
                new_e0.append(e0[a:b]); new_ln.append(ln[a:b]); new_kd.append(kd[a:b])
                new_start.append(new_start[-1] + (b - a))
            e0, ln, kd = np.concatenate(new_e0), np.concatenate(new_ln), np.concatenate(new_kd)
            wg_start = np.array(new_start, dtype=np.int64)
        out.append((e0, ln, kd, wg_start))
    stats["xcd_nnz"] = [int(o[1].sum()) for o in out]
    return out, stats


def simulate(lib, col, n_cols, sched, lines=32768, conc=160, batch=8):
    hits = np.zeros(2, np.int64)
    tot = np.zeros(2, np.int64)
    col = np.ascontiguousarray(col, np.int32)
    for e0, ln, kd, wg in sched:
        lib.simulate_xcd(col.ctypes.data, n_cols, e0.ctypes.data, ln.ctypes.data, kd.ctypes.data, wg.ctypes.data, len(wg) - 1,
                         lines, conc, batch, hits.ctypes.data, tot.ctypes.data)
    return hits, tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--thresholds", default="0,512,256,128,64,32")
    ap.add_argument("--slices", default="8,16,32")
    ap.add_argument("--lines", type=int, default=32768)
    ap.add_argument("--conc", type=int, default=160)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--interleave", action="store_true", help="sliced and free workgroups alternate instead of sliced first")
    args = ap.parse_args()
    sys.path.insert(0, R)
    import bench
    n_local, e_local, _, vw, _ = bench.WORKLOADS[args.workload]
    rp, col = bench.make_local_block(args.workload, n_local, e_local, vw, 0)
    N, E, n_cols = len(rp) - 1, len(col), n_local * vw
    lib = _lib()
    panels = max(1, args.dim // 32)
    print("%s: N=%d E=%d columns=%d; X panel = %.1f MB; L2 lines per XCD %d; %d panels" % (args.workload, N, E, n_cols, n_cols * 128 / 1e6, args.lines, panels))
    print("thr    S | hit_all hit_sliced hit_free | sliced_nnz%% partials | est. traffic/launch GB (X miss + partials w+r + Z + idx) | max/mean XCD nnz")
    for thr in [int(t) for t in args.thresholds.split(",")]:
        for S in ([0] if thr <= 0 else [int(s) for s in args.slices.split(",")]):
            t0 = time.time()
            sched, st = build_schedule(rp, col, n_cols, thr, S, interleave=args.interleave)
            hits, tot = simulate(lib, col, n_cols, sched, args.lines, args.conc)
            miss = (tot - hits).sum()
            traffic = panels * (miss * 128 + st["n_partials"] * 128 * 2 + N * 128) + 4 * E * panels + 4 * (N + 1)
            x = np.array(st["xcd_nnz"], float)
            print("%4d %4d | %.3f   %.3f      %.3f    | %5.1f %9d | %.2f | %.3f  (%.0fs)" % (
                thr, S, hits.sum() / tot.sum(), hits[1] / max(tot[1], 1), hits[0] / max(tot[0], 1),
                100.0 * st.get("sliced_nnz", 0) / E, st["n_partials"], traffic / 1e9, x.max() / x.mean(), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
