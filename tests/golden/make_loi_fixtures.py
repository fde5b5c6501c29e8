"""Generates tests/golden/loi_*.npz -- run ONLY in the authoring container.

Each fixture = a small seeded CSR graph + the vertex order and group sizes that the REFERENCE's
own LOI.cpp (compiled from /root/reference by `make -C oracle ref`, driven by
oracle/ref_loi_driver.cpp) produces for it with reorder_plus_new_direct, the variant its main
calls (LOI.cpp:848).  The loi_win_*.npz fixtures pin the two WINDOWED variants reorder_plus_direct / reorder_plus
(LOI.cpp:286-484 / :98-284) on graphs inside the domain where the reference is defined (>= 50 rows, no row without
entries); their row order comes from a non-stable std::sort, so they are tied to this container's libstdc++.
The fixtures are data (inputs + expected outputs); no reference source travels.
Usage: python tests/golden/make_loi_fixtures.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
from hcspmm import graphs  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "loi_ref")


def run_ref(rowptr, col, variant="new_direct"):
    N, E = len(rowptr) - 1, len(col)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<qq", N, E))
            f.write(np.asarray(rowptr, np.int32).tobytes())
            f.write(np.asarray(col, np.int32).tobytes())
        subprocess.check_call([REF_BIN, variant, fin, fout])
        raw = open(fout, "rb").read()
    off = 0
    (ng,) = struct.unpack_from("<q", raw, off)
    off += 8
    sizes, members = [], []
    for _ in range(ng):
        (s,) = struct.unpack_from("<i", raw, off)
        off += 4
        members.append(np.frombuffer(raw, np.int32, s, off).copy())
        off += 4 * s
        sizes.append(s)
    (no,) = struct.unpack_from("<q", raw, off)
    off += 8
    order = np.frombuffer(raw, np.int32, no, off).copy()
    return np.asarray(sizes, np.int32), np.concatenate(members) if members else np.zeros(0, np.int32), order


def cases():
    yield "powerlaw_sym_600", graphs.powerlaw_graph(600, 3000, seed=3)
    yield "powerlaw_sym_2000", graphs.powerlaw_graph(2000, 12000, seed=1)
    yield "uniform_directed_300", graphs.uniform_graph(300, 1500, seed=5)
    yield "planted_dense_400", graphs.planted_dense_graph(400, seed=7)
    # rows without out-edges + isolated vertices + N % 16 != 0
    rp, col = graphs.uniform_graph(203, 700, seed=9)
    rp = rp.copy()
    deg = np.diff(rp)
    keep = np.ones(len(col), bool)
    for r in range(0, 203, 5):
        keep[rp[r]:rp[r + 1]] = False
        deg[r] = 0
    yield "holes_203", (np.concatenate([[0], np.cumsum(deg)]).astype(np.int32), col[keep])
    yield "tiny_path_5", (np.array([0, 1, 3, 5, 7, 8], np.int32), np.array([1, 0, 2, 1, 3, 2, 4, 3], np.int32))


def _fill_empty_rows(rp, col, seed):
    """Give every row without entries one entry (the windowed variants read out of bounds on empty rows)."""
    N = len(rp) - 1
    rng = np.random.default_rng(seed)
    deg = np.diff(rp)
    rows = np.repeat(np.arange(N), deg)
    empty = np.nonzero(deg == 0)[0]
    r = np.concatenate([rows, empty]).astype(np.int64)
    c = np.concatenate([col, (empty + 1 + rng.integers(0, N - 1, len(empty))) % N]).astype(np.int64)
    rp2, col2 = graphs._to_csr(r, c, N)
    assert np.diff(rp2).min() > 0
    return rp2, col2


def windowed_cases():
    yield "uniform_64", _fill_empty_rows(*graphs.uniform_graph(64, 300, seed=11), seed=1)            # barely above the 50-row floor
    yield "powerlaw_777", _fill_empty_rows(*graphs.powerlaw_graph(777, 6000, seed=12), seed=2)       # N % 16 != 0, three windows of 300
    yield "planted_1200", _fill_empty_rows(*graphs.planted_dense_graph(1200, seed=13), seed=3)
    # many rows share their smallest column id: the order of equal keys is whatever the non-stable sort leaves
    rng = np.random.default_rng(14)
    N = 400
    rows = np.repeat(np.arange(N), 4)
    cols = np.concatenate([np.stack([rng.integers(0, 6, N), rng.integers(6, N, N), rng.integers(6, N, N), rng.integers(6, N, N)], 1).ravel()])
    yield "ties_400", graphs._to_csr(rows.astype(np.int64), cols.astype(np.int64), N)


if __name__ == "__main__":
    if not os.path.exists(REF_BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (rp, col) in cases():
        sizes, members, order = run_ref(rp, col)
        sizes_n, members_n, order_n = run_ref(rp, col, "new")  # reorder_plus_new (LOI.cpp:505-658)
        assert sorted(order.tolist()) == list(range(len(rp) - 1)), name
        np.savez_compressed(os.path.join(out_dir, "loi_%s.npz" % name), row_pointers=rp, column_index=col,
                            group_sizes=sizes, group_members=members, order=order,
                            group_sizes_new=sizes_n, group_members_new=members_n, order_new=order_n)
        print(name, "N", len(rp) - 1, "E", len(col), "groups", len(sizes), "full", int((sizes == 16).sum()))
    for name, (rp, col) in windowed_cases():
        assert np.diff(rp).min() > 0 and len(rp) - 1 >= 50
        sizes_d, members_d, order_d = run_ref(rp, col, "plus_direct")
        sizes_p, members_p, order_p = run_ref(rp, col, "plus")
        assert sorted(order_d.tolist()) == list(range(len(rp) - 1)) and sorted(order_p.tolist()) == list(range(len(rp) - 1)), name
        np.savez_compressed(os.path.join(out_dir, "loi_win_%s.npz" % name), row_pointers=rp, column_index=col,
                            group_sizes_plus_direct=sizes_d, group_members_plus_direct=members_d, order_plus_direct=order_d,
                            group_sizes_plus=sizes_p, group_members_plus=members_p, order_plus=order_p)
        print("windowed", name, "N", len(rp) - 1, "E", len(col), "groups", len(sizes_d), len(sizes_p), "full", int((sizes_d == 16).sum()))
