"""How far is a launch from the L2's own gather rate?  The same graph (same rows, same entry count, same task schedule
sizes) with its column ids folded into [0, F): at F = 2048 every gathered line is an L2 hit (256 KB per 32-column panel),
so the time that remains is the L2 -> CU path plus tasks / indices / stores -- the floor of the sparse-row path for this
degree sequence, whatever the cache behaviour of the real columns.

  python tools/gather_floor.py [--workload reddit] [--dim 128] [--folds 0,65536,2048]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--folds", default="0,65536,2048")
    ap.add_argument("--variants", default="off,256x8")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import hcspmm
    dev = torch.device("cuda:0")
    n_local, e_local, _, vw, _ = bench.WORKLOADS[args.workload]
    rp, col = bench.make_local_block(args.workload, n_local, e_local, vw, 0)
    N, E, D = len(rp) - 1, len(col), args.dim
    rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))

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

    for F in [int(f) for f in args.folds.split(",")]:
        if F > 0:
            c2 = (col.astype(np.int64) % F)
            o = np.lexsort((c2, rows))  # rows stay where they are, columns ascending inside a row (duplicates kept: same E)
            c2 = c2[o].astype(np.int32)
            M = F
        else:
            c2, M = col, n_local * vw
        rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(c2).to(dev)
        bp, e2c, e2r, ht, plan, col_nzr = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16, rule=2, num_columns=M)
        X = torch.randn(M, D, device=dev)
        Z = torch.empty(N, D, device=dev)
        for v in args.variants.split(","):
            kw = dict(slice_threshold=-1) if v == "off" else dict(slice_threshold=int(v.split("x")[0]), n_slices=int(v.split("x")[1]))
            p = hcspmm.build_plan(rp_d, col_d, bp, e2c, ht, num_columns=M, **kw)
            a = (rp_d, col_d, bp, e2c, e2r, ht, p, col_nzr)
            ws = torch.empty(max(hcspmm.workspace_bytes(p, D) // 4, 1), dtype=torch.float32, device=dev)
            t = timeit(lambda: hcspmm.forward_into(X, Z, *a, workspace=ws))
            print("%s D=%d  columns %-22s %-7s %8.1f us   gathered bytes / t = %5.2f TB/s" % (
                args.workload, D, ("folded into [0, %d)" % F) if F else "as they are (%d)" % M, v, t, E * D * 4 / t / 1e6), flush=True)
    print("Z-sized streaming fill %8.1f us" % timeit(lambda: Z.fill_(1.0)))


if __name__ == "__main__":
    main()
