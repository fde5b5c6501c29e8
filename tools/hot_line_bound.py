"""Upper bound of an LDS-resident hot-line cache for the column-sliced rows: the same launch with the entries that such a
cache would serve REMOVED from the long rows (the K most popular columns of each of the 8 slices), i.e. with LDS reads
priced at zero.  The result of that launch is wrong by construction; only its time is of interest.
  python tools/hot_line_bound.py [--k 1024] [--dim 128]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import bench, hcspmm

ap = argparse.ArgumentParser(); ap.add_argument("--k", default="512,1024,2048"); ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--thresholds", default="256,128,64")
args = ap.parse_args()
dev = torch.device("cuda:0")
rp, col = bench.make_local_block("reddit", 233000, 11600000, 1, 0)
N, D = len(rp) - 1, args.dim
deg = np.diff(rp); rows = np.repeat(np.arange(N), deg)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(tag, rp, col, thr):
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    bp, e2c, e2r, ht, _, cn = hcspmm.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16, rule=2)
    plan = hcspmm.build_plan(rp_d, col_d, bp, e2c, ht, slice_threshold=thr, n_slices=8)
    X = torch.randn(N, D, device=dev); Z = torch.empty(N, D, device=dev)
    ws = torch.empty(max(hcspmm.workspace_bytes(plan, D) // 4, 1), dtype=torch.float32, device=dev)
    t = timeit(lambda: hcspmm.forward_into(X, Z, rp_d, col_d, bp, e2c, e2r, ht, plan, cn, workspace=ws))
    print("%-58s E=%9d  %8.1f us" % (tag, len(col), t), flush=True)


for thr in [int(t) for t in args.thresholds.split(",")]:
    run("threshold %d, as it is" % thr, rp, col, thr)
    long_e = deg[rows] > thr
    c = col[long_e]
    cs = np.cumsum(np.bincount(c, minlength=N)); b = np.searchsorted(cs, cs[-1] * np.arange(1, 8) / 8)
    s_of = np.searchsorted(b, np.arange(N), side="right")
    cnt = np.bincount(c, minlength=N)
    for K in [int(k) for k in args.k.split(",")]:
        hot = np.zeros(N, bool)
        for s in range(8):
            ids = np.where(s_of == s)[0]
            hot[ids[np.argsort(-cnt[ids])[:K]]] = True
        keep = ~(long_e & hot[col])
        # sliced rows must stay "long" for the plan: thresholds are applied to the ORIGINAL degrees by keeping the split at thr
        # (a row may fall below thr after the removal: then it is simply a free row -- a slight under-estimate of the bound)
        rp2 = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=N))]).astype(np.int32)
        run("threshold %d, hot entries removed (K = %d per slice: %.1f %% of all)" % (thr, K, 100.0 * (~keep).mean()), rp2, col[keep], thr)
