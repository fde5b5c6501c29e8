"""Where does the tiny-task launch of its own start to pay?  Low-degree power-law graphs (2.1 entries per row, D = 32) of
growing size, timed in this process; run once per setting of HCSPMM_TINY_KERNEL_MIN_TASKS (-1 = never, 1 = always).
  HCSPMM_TINY_KERNEL_MIN_TASKS=-1 python tools/ab_tiny_sizes.py; HCSPMM_TINY_KERNEL_MIN_TASKS=1 python tools/ab_tiny_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm
from hcspmm import graphs
dev = torch.device("cuda:0")
for N in (150000, 300000, 600000, 1200000, 2400000):
    rp, col = graphs.powerlaw_graph(N, int(2.1 * N), seed=3)
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    h = hcspmm.plan_header(outs[4])
    for D in (32, 128):
        X = torch.randn(N, D, device=dev); Z = torch.empty(N, D, device=dev)
        ws = torch.empty(max(hcspmm.workspace_bytes(outs[4], D) // 4, 1), dtype=torch.float32, device=dev)
        f = lambda: hcspmm.forward_into(X, Z, rp_d, col_d, *outs, workspace=ws)
        for _ in range(10): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): f()
        e.record(); torch.cuda.synchronize()
        print("MIN_TASKS=%-3s N=%8d tiny tasks %8d D=%3d  %8.2f us" % (os.environ.get("HCSPMM_TINY_KERNEL_MIN_TASKS", "dflt"), N, h.n_tiny, D, s.elapsed_time(e) / 200 * 1e3), flush=True)
