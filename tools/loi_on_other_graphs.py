"""Does the (relaxed) LOI reorder help or hurt on graphs it was not made for?  SpMM time (D = 32 and 128, default classifier) before and
after hcspmm.loi_reorder(variant="fast") on the YeastH-sized molecule collection (already local), the RD- and TT-sized power-law graphs
(no structure to find) and the planted 40-column groups of the round-1 experiment (shuffled)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import bench, hcspmm
from hcspmm import graphs

dev = torch.device("cuda:0")


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


def spmm_us(rp, col):
    n = len(rp) - 1
    rp_d, col_d = torch.from_numpy(np.asarray(rp)).to(dev), torch.from_numpy(np.asarray(col)).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, n, len(col), (n + 15) // 16)
    h = hcspmm.plan_header(outs[4])
    res = []
    for D in (32, 128):
        X = torch.randn(n, D, device=dev)
        res.append(timeit(lambda: hcspmm.forward(X, rp_d, col_d, *outs)))
        del X
    torch.cuda.empty_cache()
    return res, h.n_dense


for name in sys.argv[1:] or ["yh_like", "rd_like", "tt_like", "wide_groups_shuffled"]:
    if name == "wide_groups_shuffled":
        rp, col = graphs.planted_dense_graph_fast(2000000, seed=1, dense_fraction=0.7, k_cols=40, fill=0.35, sparse_degree=14)
        sh = torch.from_numpy(np.random.default_rng(0).permutation(len(rp) - 1).astype(np.int32))
        a, b = hcspmm.apply_permutation(torch.from_numpy(rp), torch.from_numpy(col), sh)
        rp, col = a.numpy(), b.numpy()
    else:
        n, e, _, vw, _ = bench.WORKLOADS[name]
        rp, col = bench.make_local_block(name, n, e, 1, 0)
    (t32, t128), nd = spmm_us(rp, col)
    rpt, colt = torch.from_numpy(rp), torch.from_numpy(col)
    t0 = time.perf_counter()
    perm, sizes = hcspmm.loi_reorder(rpt, colt, variant="fast")
    t_loi = time.perf_counter() - t0
    a, b = hcspmm.apply_permutation(rpt, colt, perm)
    (u32, u128), nd2 = spmm_us(a.numpy(), b.numpy())
    print("%-22s %8d nodes: as given D=32 %7.1f us  D=128 %7.1f us (%6d dense windows) | after fast LOI (%.3f s, %d of %d groups full) D=32 %7.1f us (%+5.1f %%)  "
          "D=128 %7.1f us (%+5.1f %%) (%6d dense windows)" % (name, len(rp) - 1, t32, t128, nd, t_loi, int((sizes == 16).sum()), len(sizes), u32,
                                                          100 * (t32 - u32) / t32, u128, 100 * (t128 - u128) / t128, nd2), flush=True)
