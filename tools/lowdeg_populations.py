"""RD-sized low-degree graph (D = 32): time of each row population on its own -- rows of 1-2 entries (tiny tasks), 3-16,
17-256, > 256 (column-sliced) -- the other rows emptied (they still get their zero row of Z written as tiny tasks).
  python tools/lowdeg_populations.py [--workload rd_like]"""
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
deg = np.diff(rp); rows = np.repeat(np.arange(N), deg)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(tag, keep_rows):
    keep = keep_rows[rows]
    rp2 = np.concatenate([[0], np.cumsum(np.where(keep_rows, deg, 0))]).astype(np.int32)
    col2 = col[keep]
    rp_d, col_d = torch.from_numpy(rp2).to(dev), torch.from_numpy(col2).to(dev)
    outs = hcspmm.preprocess(col_d, rp_d, N, len(col2), (N + 15) // 16, rule=2)
    h = hcspmm.plan_header(outs[4])
    X = torch.randn(N, D, device=dev); Z = torch.empty(N, D, device=dev)
    ws = torch.empty(max(hcspmm.workspace_bytes(outs[4], D) // 4, 1), dtype=torch.float32, device=dev)
    t = timeit(lambda: hcspmm.forward_into(X, Z, rp_d, col_d, *outs, workspace=ws))
    print("%-34s rows %8d entries %9d (%.1f per row)  tasks %8d tiny %8d slice tasks %7d  %8.1f us" % (
        tag, int(keep_rows.sum()), len(col2), len(col2) / max(int(keep_rows.sum()), 1), h.n_tasks, h.n_tiny, h.n_slice_tasks, t), flush=True)


run("all rows", deg >= 0)
run("no entries at all (Z zero fill)", deg < 0)
run("rows of 1-2 entries", (deg >= 1) & (deg <= 2))
run("rows of 3-8 entries", (deg >= 3) & (deg <= 8))
run("rows of 9-32 entries", (deg >= 9) & (deg <= 32))
run("rows of 33-256 entries", (deg >= 33) & (deg <= 256))
run("rows of > 256 entries", deg > 256)
