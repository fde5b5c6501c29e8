"""Dense-tile path only: time per launch against the window width K (65 536 synthetic windows, fill 0.35), to see
what the compact records (K <= 32: one 256-byte load per unit) buy over the regular layout just above the limit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import hcspmm
from refit_classifier import synthetic_windows, time_us

dev = torch.device("cuda:0")
for D in (32, 128):
    for K in (16, 24, 32, 40, 48, 64, 96):
        rp, col = synthetic_windows(65536, K, 0.35, seed=K)
        N, E = len(rp) - 1, len(col)
        rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
        bp, e2c, e2r, ht, _, cn = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16, rule=2)
        ht1 = torch.full_like(ht, 1)
        plan = hcspmm.build_plan(rp_d, col_d, bp, e2c, ht1)
        X = torch.randn(N, D, device=dev)
        t = time_us(lambda: hcspmm.forward(X, rp_d, col_d, bp, e2c, e2r, ht1, plan, cn))
        gathered = 65536 * K * D * 4 + N * D * 4
        print("D=%3d K=%3d  %7.1f us   %5.2f ns/window   %.2f TB/s (unique rows + Z)" % (D, K, t, t * 1e3 / 65536, gathered / t / 1e6))
