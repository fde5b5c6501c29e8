"""Wall time of the relaxed parallel LOI reorder (hcspmm.loi_reorder(variant="fast")) against the exact one on the community-structured
RD-sized graph and on the Reddit-scale power-law graph, by thread count (host only; HCSPMM_LOI_DEBUG=1 prints the phases)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs

T = torch.from_numpy
which = sys.argv[1:] or ["community", "reddit"]
for name in which:
    if name == "community":
        rp, col, grp = graphs.community_graph(4859280, 10149830, seed=3)
    else:
        rp, col = graphs.powerlaw_graph(233000, 11600000, seed=3)
    rpt, colt = T(rp), T(col)
    print("%s: %d vertices / %d entries (host cores %d)" % (name, len(rp) - 1, len(col), os.cpu_count()), flush=True)
    ref = None
    for th in (16, 8, 4, 2, 1, 32, 0):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            perm, sizes = hcspmm.loi_reorder(rpt, colt, variant="fast", threads=th)
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = perm
        print("  fast, threads=%2d: %.3f s; %d groups, %d full; same permutation as the first run: %s"
              % (th, best, len(sizes), int((sizes == 16).sum()), bool(torch.equal(ref, perm))), flush=True)
    t0 = time.perf_counter()
    rpr, colr = hcspmm.apply_permutation(rpt, colt, perm)
    print("  apply_permutation: %.3f s" % (time.perf_counter() - t0), flush=True)
    if os.environ.get("LOI_EXACT", "1") == "1":
        t0 = time.perf_counter()
        perm, sizes = hcspmm.loi_reorder(rpt, colt)
        print("  exact (reorder_plus_new_direct, one core): %.3f s; %d groups, %d full" % (time.perf_counter() - t0, len(sizes), int((sizes == 16).sum())), flush=True)
