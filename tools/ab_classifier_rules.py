"""Which window classifier should `preprocess` use on MI355X?  SpMM time under rule 0 (the reference's coefficients, fitted on an
RTX 3090: hybrid_all_kernel.cu:261), rule 2 (as shipped: every window on the sparse-row path) and the MI355X refit for the width
(rule 3 below 64 columns, rule 4 from 64 on) on every bench.py workload that has dense-tile candidates, D = 32 and 128.
One process, HIP events, 30 calls after 5 warm-ups.  -> profiles/r04/ab_classifier_rules.log"""
import os, sys
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


def graph(name):
    if name == "community_loi":
        rp, col, _ = graphs.community_graph(4859280, 10149830, seed=3)
        rpt, colt = torch.from_numpy(rp), torch.from_numpy(col)
        perm, _ = hcspmm.loi_reorder(rpt, colt, variant="fast")
        a, b = hcspmm.apply_permutation(rpt, colt, perm)
        return a.numpy(), b.numpy(), len(rp) - 1
    if name == "wide_groups":  # the round-1 experiment: 16-row groups sharing 40 columns at 35 % fill + 30 % unstructured windows
        rp, col = graphs.planted_dense_graph_fast(2000000, seed=1, dense_fraction=0.7, k_cols=40, fill=0.35, sparse_degree=14)
        return rp, col, len(rp) - 1
    n, e, _, vw, _ = bench.WORKLOADS[name]
    rp, col = bench.make_local_block(name, n, e, vw, 0)
    return rp, col, n * vw


print("%-16s %4s | %9s %9s %9s %9s | %8s %8s %8s | refit for the width vs rule 0 | the OTHER refit vs rule 0" % (
    "workload", "D", "rule 0", "all sparse", "rule 3", "rule 4", "dense w0", "dense w3", "dense w4"))
for name in sys.argv[1:] or ["rd_like", "yh_like", "tt_like", "community", "community_loi", "wide_groups", "dense", "c5_share", "reddit"]:
    rp, col, n_cols = graph(name)
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    for D in (32, 128):
        X = torch.randn(n_cols, D, device=dev)
        res = {}
        for rule in (0, 2, 3, 4):
            outs = hcspmm.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16, rule=rule, num_columns=n_cols)
            h = hcspmm.plan_header(outs[4])
            res[rule] = (timeit(lambda: hcspmm.forward_rect(X, rp_d, col_d, *outs)), h.n_dense)
            del outs
        r = hcspmm.mi355x_rule(D)
        o = 7 - r
        print("%-16s %4d | %9.1f %9.1f %9.1f %9.1f | %8d %8d %8d | %+6.1f %% | %+6.1f %%" % (
            name, D, res[0][0], res[2][0], res[3][0], res[4][0], res[0][1], res[3][1], res[4][1],
            100.0 * (res[0][0] - res[r][0]) / res[0][0], 100.0 * (res[0][0] - res[o][0]) / res[0][0]), flush=True)
        del X
        torch.cuda.empty_cache()
    del rp_d, col_d
