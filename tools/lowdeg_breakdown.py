"""Where does a low-degree launch (RD-sized, D=32) spend its time?  Same row-degree sequence, three
column distributions: (a) the real one (X = 622 MB, random 128-B lines from HBM), (b) columns folded into
[0, 4096) (X = 512 KB: every gather is an L2 hit; what remains is tasks + indices + the Z stores),
(c) as (b) but Z rows written to a single 4096-row buffer is not possible -- so (b) bounds the non-gather cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs

dev = torch.device("cuda:0")
rp, col = graphs.powerlaw_graph(4859280, 10149830, seed=3)
N, D = len(rp) - 1, 32


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(tag, rp, col, xrows):
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16, rule=2)
    X = torch.randn(xrows, D, device=dev)
    t = timeit(lambda: hcspmm.forward_rect(X, rp_d, col_d, *outs))
    print("%-28s E=%d  %.1f us" % (tag, len(col), t))


run("real columns (X 622 MB)", rp, col, N)
# fold columns into [0, 4096), keep rows sorted and duplicate-free
rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
key = np.unique(rows * 4096 + (col.astype(np.int64) % 4096))
rows2, col2 = key // 4096, (key % 4096).astype(np.int32)
rp2 = np.zeros(N + 1, np.int64); np.add.at(rp2, rows2 + 1, 1); rp2 = np.cumsum(rp2).astype(np.int32)
run("columns folded (X 512 KB)", rp2, col2, 4096)
Z = torch.empty(N, D, device=dev)
print("Z-sized streaming fill           %.1f us" % timeit(lambda: Z.fill_(1.0)))
