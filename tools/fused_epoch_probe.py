"""Host and device cost of one fused call inside a stream of calls (extension front-end, as the GNN layers call it):
host issue time per call and wall per call with one sync at the end, for contiguous and transposed-view weights.
  HCSPMM_FUSED_SINGLE_LAUNCH=0|2 python tools/fused_epoch_probe.py rd_like"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd"), os.path.join(ROOT, "hc-spmm_amd", "hybrid_kernel")]
import torch
import bench
import HCSPMM

dev = torch.device("cuda:0")
wl = sys.argv[1]
n, e = bench.WORKLOADS[wl][0], bench.WORKLOADS[wl][1]
rp, col = bench.make_local_block(wl, n, e, 1, 0)
rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
outs = HCSPMM.preprocess(col_d, rp_d, n, len(col), (n + 15) // 16)
X = torch.randn(n, 32, device=dev)
W = torch.randn(32, 32, device=dev)
for name, w in (("W", W), ("W^T view", W.t())):
    for _ in range(10):
        HCSPMM.forward_fixed32_fused(X, rp_d, col_d, *outs, w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        HCSPMM.forward_fixed32_fused(X, rp_d, col_d, *outs, w)
    th = (time.perf_counter() - t0) / 100 * 1e6
    torch.cuda.synchronize()
    print("%s fused (32,32) %s: host issue %.1f us/call, wall %.1f us/call, form %d" % (
        wl, name, th, (time.perf_counter() - t0) / 100 * 1e6, HCSPMM.fused_in_launch(outs[4], 32, 32)))
