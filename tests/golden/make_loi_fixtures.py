"""Generates tests/golden/loi_*.npz -- run ONLY in the authoring container.

Each fixture = a small seeded CSR graph + the vertex order and group sizes that the REFERENCE's
own LOI.cpp (compiled from /root/reference by `make -C oracle ref`, driven by
oracle/ref_loi_driver.cpp) produces for it with reorder_plus_new_direct, the variant its main
calls (LOI.cpp:848).  The fixtures are data (inputs + expected outputs); no reference source
travels.  Usage: python tests/golden/make_loi_fixtures.py
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
