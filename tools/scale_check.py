"""One-off capacity check at BASELINE config 5's FULL size on one GPU: 16 M nodes / 256 M stored
entries, D = 128 (X and Z 8.2 GB each, N*D = 2.05e9 > 2^31); `scale_check.py N E D` for other sizes
(config 4: 2.45e6 6.2e7 256).  Verifies exact integer checksums
(X = 1 -> degrees; X[i,:] = i mod 251), prints preprocess and SpMM times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import hcspmm

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
E = int(float(sys.argv[2])) if len(sys.argv) > 2 else 256_000_000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
t0 = time.perf_counter()
rng = np.random.default_rng(7)
# degree-skewed rows (power law on the row side), uniformly random columns; duplicates merged by sort
w = (np.arange(N, dtype=np.float64) + 1.0) ** (-0.75)
w = np.minimum(w / w.sum(), 5e-6)
cdf = np.cumsum(w / w.sum()); cdf[-1] = 1.0
perm = rng.permutation(N)
rows = perm[np.searchsorted(cdf, rng.random(E))].astype(np.int64)
cols = rng.integers(0, N, E, dtype=np.int64)
key = np.unique(rows * N + cols)
del rows, cols
rows, cols = key // N, (key % N).astype(np.int32)
del key
rp = np.zeros(N + 1, np.int64)
np.add.at(rp, rows + 1, 1)
rp = np.cumsum(rp).astype(np.int32)
del rows
E = len(cols)
print("graph: N=%d E=%d max degree %d (%.0f s)" % (N, E, int(np.diff(rp).max()), time.perf_counter() - t0), flush=True)
rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(cols).to(dev)
t0 = time.perf_counter()
outs = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16)
torch.cuda.synchronize()
h = hcspmm.plan_header(outs[4])
print("preprocess %.2f s: tasks %d dense windows %d split rows %d plan %.1f MB" % (
    time.perf_counter() - t0, h.n_tasks, h.n_dense, h.n_split_rows, h.total_words * 4 / 1e6), flush=True)
a = (rp_d, col_d, *outs)
X = torch.ones(N, D, device=dev)
Z = hcspmm.forward(X, *a)[0]
deg = torch.from_numpy(np.diff(rp).astype(np.float32)).to(dev)
assert torch.equal(Z[:, 0], deg) and torch.equal(Z[:, D - 1], deg)
ids = (torch.arange(N, device=dev) % 251).float()
X = ids[:, None].expand(N, D).contiguous()
Z = hcspmm.forward(X, *a)[0]
cs = np.concatenate([[0.0], np.cumsum((cols % 251).astype(np.float64))])
want = torch.from_numpy((cs[rp[1:]] - cs[rp[:-1]]).astype(np.float32)).to(dev)
assert torch.equal(Z[:, 0], want) and torch.equal(Z[:, 77], want)
print("exact integer checksums OK (N*D = %.3g elements)" % (N * D), flush=True)
X = torch.randn(N, D, device=dev)
for _ in range(3):
    hcspmm.forward(X, *a)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    hcspmm.forward(X, *a)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
b_alg = 4.0 * E * D + 4.0 * N * D + 4.0 * E + 4.0 * (N + 1)
print("SpMM %.2f ms  = %.3g edge*dim/s, %.2f TB/s algorithmic" % (ms, E * D / ms * 1e3, b_alg / ms / 1e9), flush=True)
