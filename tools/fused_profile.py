import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import torch, bench, hcspmm
dev = torch.device("cuda:0")
wl = sys.argv[1]; D, H = int(sys.argv[2]), int(sys.argv[3])
n_local, e_local, _, vw, _ = bench.WORKLOADS[wl]
rp, col = bench.make_local_block(wl, n_local, e_local, 1, 0)
N, E = len(rp) - 1, len(col)
rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
outs = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16)
X, W = torch.randn(N, D, device=dev), torch.randn(D, H, device=dev)
for _ in range(12):
    hcspmm.forward_fixed32_fused(X, rp_d, col_d, *outs, W)
torch.cuda.synchronize()
