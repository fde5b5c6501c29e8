#!/usr/bin/env python3
"""Refit of the sparse-vs-dense window classifier for MI355X.

The reference's coefficients (hybrid_all_kernel.cu:261: 0.19854024, -6.578043, -3.14922857) were fit
on an RTX 3090 with the paper's procedure (p.6-7): synthetic 16 x K windows, K = 1..130, time both
sub-paths, logistic regression on  x1 = K - 1,  x2 = nnz / (ceil(K/8) * 128)  with label
"sparse-row path is faster".  This tool repeats that procedure on the GPU it runs on with OUR two
sub-paths (planned kernel, every window forced onto one path via hcspmm.build_plan) and prints the
measurements and the fitted coefficients as JSON.

  python tools/refit_classifier.py [--dims 32 128] [--windows 16384] > profiles/rNN/classifier_refit.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hcspmm  # noqa: E402


def synthetic_windows(n_windows, K, fill, seed):
    """n_windows 16-row windows, each with exactly K distinct columns; every column has >= 1 entry,
    further cells are set with probability `fill`."""
    rng = np.random.default_rng(seed)
    N = n_windows * 16
    cols = np.stack([rng.choice(N, K, replace=False) for _ in range(min(n_windows, 512))])
    cols = cols[rng.integers(0, cols.shape[0], n_windows)]                 # [W, K]
    mask = rng.random((n_windows, 16, K)) < fill
    first = rng.integers(0, 16, (n_windows, K))
    mask[np.arange(n_windows)[:, None], first, np.arange(K)[None, :]] = True
    w, r, k = np.nonzero(mask)
    rows = w * 16 + r
    c = cols[w, k]
    order = np.lexsort((c, rows))
    rows, c = rows[order], c[order]
    rp = np.zeros(N + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp).astype(np.int32), c.astype(np.int32)


def time_us(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[32, 128])
    ap.add_argument("--windows", type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    Ks = [1, 2, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 112, 130]
    fills = [0.0, 0.05, 0.1, 0.2, 0.35, 0.5, 0.75, 1.0]
    rows = []
    for K in Ks:
        for fill in fills:
            rp, col = synthetic_windows(args.windows, K, fill, seed=K * 100 + int(fill * 100))
            N, E = len(rp) - 1, len(col)
            rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
            bp, e2c, e2r, ht, _, cn = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16, rule=2)  # all sparse
            plans = {}
            for t in (0, 1):
                plans[t] = hcspmm.build_plan(rp_d, col_d, bp, e2c, torch.full_like(ht, t))
            num = (K - 1 + 8) // 8
            for D in args.dims:
                X = torch.randn(N, D, device=dev)
                tt = {}
                for t in (0, 1):
                    htt = torch.full_like(ht, t)
                    tt[t] = time_us(lambda: hcspmm.forward(X, rp_d, col_d, bp, e2c, e2r, htt, plans[t], cn))
                rows.append({"K": K, "fill": fill, "D": D, "nnz_per_window": E / args.windows,
                             "x1": K - 1, "x2": (E / args.windows) / (num * 128.0),
                             "sparse_us": tt[0], "dense_us": tt[1], "sparse_faster": bool(tt[0] < tt[1])})
                print("K=%3d fill=%.2f D=%3d nnz/w=%7.1f sparse %8.1f us dense %8.1f us" % (
                    K, fill, D, E / args.windows, tt[0], tt[1]), file=sys.stderr)
    out = {"device": torch.cuda.get_device_name(0), "windows": args.windows, "measurements": rows, "fits": {}}
    try:
        from sklearn.linear_model import LogisticRegression
        for D in args.dims + ["all"]:
            sel = [r for r in rows if D == "all" or r["D"] == D]
            Xf = np.array([[r["x1"], r["x2"]] for r in sel])
            y = np.array([r["sparse_faster"] for r in sel], int)
            if y.min() == y.max():
                out["fits"][str(D)] = {"note": "one class only: %s always faster" % ("sparse" if y[0] else "dense")}
                continue
            lr = LogisticRegression(C=1e4, max_iter=10000).fit(Xf, y)
            acc = float((lr.predict(Xf) == y).mean())
            ref = (0.19854024 * Xf[:, 0] - 6.578043 * Xf[:, 1] - 3.14922857) > 0
            out["fits"][str(D)] = {"w1": float(lr.coef_[0][0]), "w2": float(lr.coef_[0][1]), "b": float(lr.intercept_[0]),
                                   "accuracy": acc, "reference_coefficients_accuracy": float((ref == y.astype(bool)).mean()),
                                   "sparse_faster_fraction": float(y.mean())}
    except Exception as e:  # sklearn missing: measurements are still useful
        out["fits"]["error"] = str(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
