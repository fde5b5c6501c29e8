#!/usr/bin/env python3
"""Layout reorder as a file-to-file program -- counterpart of the reference's LOI.cpp `main` (LOI.cpp:807-896), which reads a graph
in the dataset text format, runs reorder_plus_new_direct and writes the new vertex order to reorder_direct.txt, one 0-based old
vertex id per line (full 16-row groups first, then the short groups, then the vertices without out-edges), printing the run time and
the number of full groups.  The reference hard-codes the file name and the node / edge counts and leaves applying the order to the
user; here they are arguments, and --apply writes the relabelled graph back in the same text format.

  python LOI.py Dataset/example.txt                         # -> reorder_direct.txt, the reference's order bit for bit
  python LOI.py Dataset/example.txt --variant fast --apply Dataset/example_loi.txt

The text format is the reference's (readCSR, LOI.cpp:486-503; dataset.py:52-53): one "a,b" line per stored entry, 1-based, sorted by
b -- row b-1 holds column a-1.  Lines are taken as they come (readCSR merges nothing): a file with duplicate lines keeps them.
Host only: no GPU is touched.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import hcspmm  # noqa: E402  (ctypes front-end of libhcspmm.so: the reorder is a host routine of the library)


def read_csr(path):
    """readCSR (LOI.cpp:486-503): the second field is the 1-based row, ascending; the first the 1-based column."""
    raw = np.loadtxt(path, delimiter=",", dtype=np.int64, ndmin=2)
    if raw.size == 0:
        return np.zeros(1, np.int32), np.zeros(0, np.int32)
    a, b = raw[:, 0], raw[:, 1]
    if np.any(np.diff(b) < 0):
        raise SystemExit("LOI.py: %s is not sorted by its second field (the reference's reader needs that, LOI.cpp:493-499)" % path)
    n = int(max(a.max(), b.max()))
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, b, 1)  # b is 1-based: counts land at rowptr[row + 1]
    return np.cumsum(rowptr).astype(np.int32), (a - 1).astype(np.int32)


def write_graph(path, rowptr, col):
    with open(path, "w") as f:
        for r in range(len(rowptr) - 1):
            for e in range(rowptr[r], rowptr[r + 1]):
                f.write("%d,%d\n" % (col[e] + 1, r + 1))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("graph", help='graph file in the dataset text format ("a,b" per line, 1-based, sorted by b)')
    ap.add_argument("--out", default="reorder_direct.txt", help="where the order goes (the reference's file name by default)")
    ap.add_argument("--variant", default="new_direct", choices=["new_direct", "new", "plus_direct", "plus", "fast"],
                    help="new_direct = reorder_plus_new_direct, the one the reference's main calls; fast = the relaxed parallel variant "
                         "(not the reference's order)")
    ap.add_argument("--apply", default="", help="also write the relabelled graph to this file (same text format)")
    args = ap.parse_args(argv)
    rowptr, col = read_csr(args.graph)
    rpt, colt = torch.from_numpy(rowptr), torch.from_numpy(col)
    start = time.perf_counter()
    perm, sizes = hcspmm.loi_reorder(rpt, colt, variant=args.variant)
    print("All time: %gs" % (time.perf_counter() - start))  # LOI.cpp:850-852
    np.savetxt(args.out, perm.numpy(), fmt="%d")
    print(int((sizes == 16).sum()))  # LOI.cpp:892: the number of full groups
    if args.apply:
        rp2, col2 = hcspmm.apply_permutation(rpt, colt, perm)
        write_graph(args.apply, rp2.numpy(), col2.numpy())
    return perm, sizes


if __name__ == "__main__":
    main()
