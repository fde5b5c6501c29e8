"""Experiment behind bench.py's `loi` block: the community-structured RD-sized graph (hcspmm.graphs.community_graph) as it arrives
(vertex ids shuffled), after the relaxed parallel LOI reorder and after the exact one; for each the windows on the dense-tile path
and the SpMM time under the reference's classifier (rule 0) and the MI355X refit, D = 32 and 128; plus the reorder wall times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4859280
E = int(sys.argv[2]) if len(sys.argv) > 2 else 10149830
EXACT = os.environ.get("LOI_EXACT", "1") == "1"


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
    for D in (32, 128):
        X = torch.randn(n, D, device=dev)
        for rule in (0, hcspmm.mi355x_rule(D), 2):
            outs = hcspmm.preprocess(col_d, rp_d, n, len(col), (n + 15) // 16, rule=rule)
            h = hcspmm.plan_header(outs[4])
            t = timeit(lambda: hcspmm.forward(X, rp_d, col_d, *outs))
            print("%-30s D=%3d rule %d: dense windows %6d / %6d  entries on dense path %4.1f %%  uniq_dense %8d  %8.1f us"
                  % (tag, D, rule, h.n_dense, (n + 15) // 16, 100.0 * h.nnz_dense / max(len(col), 1), h.uniq_dense, t), flush=True)
            del outs
        del X
        torch.cuda.empty_cache()


t0 = time.perf_counter()
rp, col, grp = graphs.community_graph(N, E, seed=3)
print("community_graph: %d vertices / %d entries, generated in %.1f s" % (N, len(col), time.perf_counter() - t0), flush=True)
measure("vertex ids shuffled", rp, col)
rpt, colt = torch.from_numpy(rp), torch.from_numpy(col)
for th in (16, 8, 4, 32):
    t0 = time.perf_counter()
    perm, sizes = hcspmm.loi_reorder(rpt, colt, variant="fast", threads=th)
    print("fast LOI, %2d threads: %.3f s; %d groups, %d full" % (th, time.perf_counter() - t0, len(sizes), int((sizes == 16).sum())), flush=True)
t0 = time.perf_counter()
perm, sizes = hcspmm.loi_reorder(rpt, colt, variant="fast")
t_fast = time.perf_counter() - t0
t0 = time.perf_counter()
rpr, colr = hcspmm.apply_permutation(rpt, colt, perm)
print("fast LOI (automatic): %.3f s + apply_permutation %.3f s" % (t_fast, time.perf_counter() - t0), flush=True)
measure("after fast LOI", rpr.numpy(), colr.numpy())
if EXACT:
    t0 = time.perf_counter()
    perm, sizes = hcspmm.loi_reorder(rpt, colt)
    print("exact LOI (reorder_plus_new_direct, one core): %.3f s; %d groups, %d full" % (time.perf_counter() - t0, len(sizes), int((sizes == 16).sum())), flush=True)
    rpr, colr = hcspmm.apply_permutation(rpt, colt, perm)
    measure("after exact LOI", rpr.numpy(), colr.numpy())
inv = np.argsort(grp, kind="stable").astype(np.int32)  # the planted numbering: groups consecutive
rpr, colr = hcspmm.apply_permutation(rpt, colt, torch.from_numpy(inv))
measure("planted order (groups consecutive)", rpr.numpy(), colr.numpy())
