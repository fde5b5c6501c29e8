"""A/B of the XCD-affine column slices (include/hcspmm.h n_slices) on one GPU: the same graph and X,
plans built with different (slice_threshold, n_slices, segment_len), each checked against the unsliced launch and timed
with HIP events.  L2 hit rate / fabric traffic of a chosen variant: bench.py under HCSPMM_SLICE_THRESHOLD / HCSPMM_SLICES
(its counter passes inherit the environment).

  python tools/ab_slices.py [--workloads reddit:128,reddit:32,reddit:256,products_share:256,c5_share:128]
                            [--variants off,64x8,128x8,32x8,64x16] [--steps 100]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="reddit:128,reddit:32,reddit:256,products_share:256,c5_share:128")
    ap.add_argument("--variants", default="off,512x8,128x8,64x8,32x8,64x16")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import hcspmm
    dev = torch.device("cuda:0")
    for spec in args.workloads.split(","):
        t0 = time.time()
        if spec.startswith("pl:"):  # pl:N:E:D -- a square power-law graph of any size (mid-size points between configs 2 and 3)
            _, n_local, e_local, D = spec.split(":")
            wl, n_local, e_local, D, vw = spec, int(n_local), int(e_local), int(D), 1
            from hcspmm import graphs
            rp, col = graphs.powerlaw_graph(n_local, e_local, seed=3)
        else:
            wl, D = spec.split(":")
            D = int(D)
            n_local, e_local, _, vw, _ = bench.WORKLOADS[wl]
            rp, col = bench.make_local_block(wl, n_local, e_local, vw, 0)
        N, E, M = len(rp) - 1, len(col), n_local * vw
        rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
        bp, e2c, e2r, ht, plan0, col_nzr = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16, num_columns=M)
        torch.manual_seed(1)
        X = torch.randn(M, D, device=dev)
        absX = X.abs()
        print("== %s D=%d: N=%d E=%d columns=%d (graph %.0fs)" % (wl, D, N, E, M, time.time() - t0), flush=True)
        ref = None
        for v in args.variants.split(","):
            if v == "auto":
                kw = {}
            elif v == "off":
                kw = dict(slice_threshold=-1)
            else:
                parts = v.split("x")
                kw = dict(slice_threshold=int(parts[0]), n_slices=int(parts[1]))
                if len(parts) > 2:
                    kw["segment_len"] = int(parts[2])
                    kw["split_threshold"] = max(512, int(parts[2]))
            plan = hcspmm.build_plan(rp_d, col_d, bp, e2c, ht, num_columns=M, **kw)
            h = hcspmm.plan_header(plan)
            a = (rp_d, col_d, bp, e2c, e2r, ht, plan, col_nzr)
            Z = torch.empty(N, D, device=dev)
            ws = torch.empty(max(hcspmm.workspace_bytes(plan, D) // 4, 1), dtype=torch.float32, device=dev)
            hcspmm.forward_into(X, Z, *a, workspace=ws)
            torch.cuda.synchronize()
            if ref is None:
                ref = Z.clone()
                # |A| * |X|: the scale of the 1e-5 bar (tests/test_spmm_gpu.py)
                Zabs = torch.empty(N, D, device=dev)
                hcspmm.forward_into(absX, Zabs, *a, workspace=ws)
                torch.cuda.synchronize()
                err = 0.0
            else:
                err = float(((Z - ref).abs() / (1e-5 * Zabs + 1e-30)).max())
            for _ in range(10):
                hcspmm.forward_into(X, Z, *a, workspace=ws)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.steps):
                hcspmm.forward_into(X, Z, *a, workspace=ws)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.steps
            print("  %-10s %8.4f ms  %6.2fe12 edge*dim/s | sliced rows %7d nnz %5.1f %% partials %8d (%6.1f MB) slice tasks %8d | "
                  "max |dZ| / (1e-5 |A||X|) = %.3f" % (v, ms, E * D / ms / 1e9, h.n_sliced_rows, 100.0 * h.nnz_sliced / max(E, 1),
                                                       h.n_partials, h.n_partials * D * 4 / 1e6, h.n_slice_tasks, err), flush=True)
            assert err <= 1.0, "sliced result off the 1e-5 bar"
            del plan, Z, ws
        del X, absX, ref, Zabs, rp_d, col_d, bp, e2c, e2r, ht, plan0
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
