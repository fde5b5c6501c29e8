"""Experiment: what the layout reorder (LOI) and the dense-tile path buy on MI355X for a graph that HAS
window structure but arrives with shuffled vertex ids.  For each stage prints windows on the
dense-tile path and the SpMM time (D = 128 and 32), for the reference's rule (0) and the MI355X refits (3 narrow, 4 wide)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def measure(tag, rp, col):
    n = len(rp) - 1
    rp_d, col_d = torch.from_numpy(np.asarray(rp)).to(dev), torch.from_numpy(np.asarray(col)).to(dev)
    for rule in (0, 3, 4):
        outs = hcspmm.preprocess(col_d, rp_d, n, len(col), (n + 15) // 16, rule=rule)
        h = hcspmm.plan_header(outs[4])
        ts = []
        for D in (128, 32):
            X = torch.randn(n, D, device=dev)
            ts.append(timeit(lambda: hcspmm.forward(X, rp_d, col_d, *outs)))
        print("%-28s rule %d: dense windows %6d / %6d  entries on dense path %4.1f %%   D=128 %7.1f us   D=32 %6.1f us"
              % (tag, rule, h.n_dense, (n + 15) // 16, 100.0 * h.nnz_dense / max(len(col), 1), ts[0], ts[1]))


# community-structured graph: 16-row groups sharing 40 columns at 35 % fill (too wide / too sparse for
# the 3090-fit rule, squarely inside the MI355X refit's dense region), plus 30 % unstructured windows
rp, col = graphs.planted_dense_graph_fast(N, seed=1, dense_fraction=0.7, k_cols=40, fill=0.35, sparse_degree=14)
measure("as generated (grouped)", rp, col)
shuffle = torch.from_numpy(np.random.default_rng(0).permutation(N).astype(np.int32))
rps, cols = hcspmm.apply_permutation(torch.from_numpy(rp), torch.from_numpy(col), shuffle)
measure("vertex ids shuffled", rps.numpy(), cols.numpy())
t0 = time.perf_counter()
perm, sizes = hcspmm.loi_reorder(rps, cols)
t_loi = time.perf_counter() - t0
rpr, colr = hcspmm.apply_permutation(rps, cols, perm)
measure("after LOI reorder", rpr.numpy(), colr.numpy())
print("LOI reorder (host, 1 core): %.2f s for %d vertices / %d entries; %d groups, %d full" % (
    t_loi, N, len(col), len(sizes), int((sizes == 16).sum())))
