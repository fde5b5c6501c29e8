"""A/B of the fused aggregate+update operators (reference hybrid_all_kernel.cu:1639-1848 etc.):
  two-launch form  : hybrid SpMM launch (writes out2 = A*X) + streaming MFMA update launch over ALL rows;
  in-launch form   : dense-tile windows multiply their tile by W inside the hybrid launch (tile kept in the MFMA
                     accumulators), update launch restricted to the sparse-row windows;
  row-tile form    : the sparse-row path as well -- tiles of 16 tasks summed, parked in LDS and multiplied in one launch
                     (fused_rows.hip), a small update launch for the rows summed by whole waves or in pieces.
Each form runs in its own process (HCSPMM_FUSED_SINGLE_LAUNCH = 0 / 1 / 2 forces the form for every plan; read once).

  python tools/ab_fused.py            -> prints one table; the builder keeps it as profiles/r02/ab_fused.log
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]

WORKLOADS = ["dense", "yh_like", "c5_share", "alldense", "reddit"]
SHAPES = [(32, 32), (128, 32), (64, 64)]
if os.environ.get("AB_SHAPES"):  # e.g. AB_SHAPES=32x32,128x32
    SHAPES = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["AB_SHAPES"].split(",")]


def child():
    import numpy as np
    import torch
    import bench
    import hcspmm
    dev = torch.device("cuda:0")

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
    out = {}
    for wl in sys.argv[2].split(","):
        n_local, e_local, _, vw, _ = bench.WORKLOADS[wl]
        scale = float(os.environ.get("AB_SCALE", 1.0))  # the same generator at a fraction of the size (crossover of the forms)
        n_local, e_local = int(n_local * scale) // 16 * 16, int(e_local * scale)
        if vw != 1:  # the fused operators are square (out2 has the graph's rows): use the block's own columns only
            from hcspmm import graphs
            rp, col = graphs.planted_powerlaw_block(n_local, n_local, e_local, seed=3)
        else:
            rp, col = bench.make_local_block(wl, n_local, e_local, 1, 0)
        N, E = len(rp) - 1, len(col)
        rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
        outs = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16)
        h = hcspmm.plan_header(outs[4])
        for D, H in SHAPES:
            X, W = torch.randn(N, D, device=dev), torch.randn(D, H, device=dev)
            a = (rp_d, col_d, *outs)
            t_sp = timeit(lambda: hcspmm.forward(X, *a))
            t_fu = timeit(lambda: hcspmm.forward_fixed32_fused(X, *a, W))
            out["%s D=%d H=%d" % (wl, D, H)] = {"spmm_us": t_sp, "fused_us": t_fu, "dense_windows": h.n_dense,
                                                 "sparse_windows": h.n_sparse_windows,
                                                 "in_launch": hcspmm.fused_in_launch(outs[4], D, H)}
        del outs, rp_d, col_d
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))


def main():
    wls = sys.argv[1] if len(sys.argv) > 1 else ",".join(WORKLOADS)
    res = {}
    for mode in ("0", "1", "2"):
        env = dict(os.environ, HCSPMM_FUSED_SINGLE_LAUNCH=mode)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wls], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-2000:])
            sys.exit(1)
        res[mode] = json.loads(line[0][7:])
    print("fused aggregate+update, one MI355X; times in us per call (50 calls after 5 warm-ups, HIP events)")
    print("forms: two launches | dense windows in the hybrid launch (1) | sparse rows as well: row tiles (2); (n) = form actually taken")
    print("%-26s %9s %9s | %12s %16s %16s | %s" % ("workload / shape", "dense win", "sparse win", "two launches", "in-launch", "row tiles",
                                                   "SpMM alone"))
    for k in res["0"]:
        a, b, c = res["0"][k], res["1"][k], res["2"][k]
        gain = lambda x: 100.0 * (a["fused_us"] - x["fused_us"]) / a["fused_us"]
        print("%-26s %9d %9d | %12.1f %8.1f %+5.1f%% (%d) %8.1f %+5.1f%% (%d) | %10.1f" % (
            k, a["dense_windows"], a["sparse_windows"], a["fused_us"], b["fused_us"], gain(b), int(b["in_launch"]), c["fused_us"], gain(c),
            int(c["in_launch"]), a["spmm_us"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
