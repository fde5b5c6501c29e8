"""Low-degree launch (RD-sized, D = 32), where does the time go?  (a) as it is, (b) columns folded into [0, 2048)
(every gather an L2 hit), (c) the same rows with NO entries (descriptor reads + Z stores only), (d) a Z-sized fill.
  python tools/lowdeg_floor.py [--workload rd_like]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import bench, hcspmm

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="rd_like"); ap.add_argument("--dim", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
n_local, e_local, _, vw, _ = bench.WORKLOADS[args.workload]
rp, col = bench.make_local_block(args.workload, n_local, e_local, vw, 0)
N, D = len(rp) - 1, args.dim


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(tag, rp, col, M):
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16, rule=2, num_columns=M)
    h = hcspmm.plan_header(outs[4])
    X = torch.randn(M, D, device=dev); Z = torch.empty(N, D, device=dev)
    ws = torch.empty(max(hcspmm.workspace_bytes(outs[4], D) // 4, 1), dtype=torch.float32, device=dev)
    t = timeit(lambda: hcspmm.forward_into(X, Z, rp_d, col_d, *outs, workspace=ws))
    print("%-44s E=%9d tasks %8d tiny %8d slices %d  %8.1f us" % (tag, len(col), h.n_tasks, h.n_tiny, h.n_slices, t), flush=True)


run("as it is", rp, col, n_local * vw)
rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rp))
c2 = col.astype(np.int64) % 2048
o = np.lexsort((c2, rows))
run("columns folded into [0, 2048): all L2 hits", rp, c2[o].astype(np.int32), 2048)
run("no entries: descriptors + Z stores only", np.zeros(N + 1, np.int32), np.zeros(0, np.int32), 2048)
Z = torch.empty(N, D, device=dev)
print("Z-sized streaming fill %8.1f us" % timeit(lambda: Z.fill_(1.0)))
