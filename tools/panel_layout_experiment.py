"""Experiment: is it worth re-laying X out panel-major ([D/32][N][32]) before the product?
Times (a) the operator as is, (b) one re-layout copy + D/32 strided products on contiguous panels."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs
dev = torch.device("cuda:0")
rp, col = graphs.powerlaw_graph(233000, 11600000, seed=3)
N, E = len(rp) - 1, len(col)
rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
outs = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16)
a = (rp_d, col_d, *outs)


def timeit(fn, n=50):
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


for D in (64, 128, 256):
    X = torch.randn(N, D, device=dev)
    Z = torch.empty(N, D, device=dev)
    P = D // 32

    def relayout():
        Xp = X.view(N, P, 32).permute(1, 0, 2).contiguous()
        for p in range(P):
            hcspmm.forward_into(Xp[p], Z[:, 32 * p:32 * p + 32], *a)
        return Z

    t_a = timeit(lambda: hcspmm.forward(X, *a))
    t_b = timeit(relayout)
    t_copy = timeit(lambda: X.view(N, P, 32).permute(1, 0, 2).contiguous())
    ok = torch.equal(relayout(), hcspmm.forward(X, *a)[0])
    print("D=%d  as-is %.1f us   relayout+%d panel products %.1f us (copy alone %.1f us)  same bits: %s" % (D, t_a, P, t_b, t_copy, ok))
