#!/usr/bin/env python3
"""Randomised parity soak on the GPU: random graphs (sizes, degree laws, hubs, empty rows), embedding widths,
feature types, classifier rules / forced sub-paths, plan tunables (tiny split thresholds so that segments,
fix-ups and tiny segments appear everywhere), the plan-free kernel, BOTH front-ends (ctypes glue / HCSPMM extension)
and the fused operators in both forms (two launches / dense windows updated in the launch) -- every result checked
against the CPU oracle with the criteria of tests/test_spmm_gpu.py.  Test infrastructure (it imports the oracle, so it lives under
tests/), but not collected by pytest -- it costs minutes of GPU time:

  python tests/fuzz_parity.py [--cases 300] [--seed 1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd"), os.path.join(ROOT, "tests")]  # ROOT = the repository
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hcspmm  # noqa: E402
import oracle  # noqa: E402
from hcspmm import graphs  # noqa: E402
import frontends  # noqa: E402
import test_spmm_gpu as T  # noqa: E402  (Graph, _check, _check_h16)


def random_graph(rng):
    kind = rng.integers(0, 6)
    N = int(rng.choice([1, 2, 15, 16, 17, 100, 1000, 3000, 9000][:]) if rng.random() < 0.3 else rng.integers(1, 6000))
    if kind == 0:
        rp, col = graphs.powerlaw_graph(max(N, 8), max(N, 8) * int(rng.integers(1, 30)), seed=int(rng.integers(1 << 30)),
                                        max_degree_frac=float(rng.choice([0.02, 0.3, 0.9])), symmetric=bool(rng.integers(0, 2)))
    elif kind == 1:
        rp, col = graphs.uniform_graph(max(N, 4), max(N, 4) * int(rng.integers(0, 12)) + 1, seed=int(rng.integers(1 << 30)))
    elif kind == 2:
        rp, col = graphs.planted_dense_graph_fast(max(N, 32), seed=int(rng.integers(1 << 30)), dense_fraction=float(rng.random()),
                                                  k_cols=int(rng.choice([4, 12, 20, 33, 40, 41, 64])), fill=float(rng.uniform(0.1, 0.9)),
                                                  sparse_degree=int(rng.integers(1, 20)))
    elif kind == 3:
        rp, col = graphs.molecule_graph(max(N, 50), seed=int(rng.integers(1 << 30)))
    elif kind == 4:  # explicit degree list: empty rows, rows of 1..3, a few hubs
        n = max(N, 3)
        deg = rng.choice([0, 0, 1, 2, 3, 5, 17, 64, 65, 130], size=n)
        deg = np.minimum(deg, n - 1)
        if n > 600:
            deg[rng.integers(0, n, 3)] = rng.integers(n // 3, n - 1, 3)
        rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        col = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg if d] or [np.zeros(0, np.int64)]).astype(np.int32)
    else:
        n = max(N, 1)
        rp, col = np.zeros(n + 1, np.int32), np.zeros(0, np.int32)  # no entries at all
    return rp, col


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    stats = {}
    for case in range(args.cases):
        rp, col = random_graph(rng)
        N = len(rp) - 1
        D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 20, 22, 31, 32, 33, 40, 64, 70, 96, 100, 128, 130, 256, 260]))
        mode = str(rng.choice(["rule0", "rule2", "rule3", "rule4", "all_dense", "all_sparse", "plan_free", "tiny_splits", "slices", "slices"]))
        dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
        rule = {"rule2": 2, "rule3": 3, "rule4": 4}.get(mode, 0)
        fe = frontends.get(["ctypes", "extension"][int(rng.integers(0, 2))])
        g = T.Graph(rp, col, dev, rule=rule, plan=(mode != "plan_free"), force_type={"all_dense": 1, "all_sparse": 0}.get(mode), fe=fe)
        if mode == "tiny_splits":  # rows longer than 5 entries are cut into segments of 1..5
            g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, split_threshold=int(rng.integers(2, 6)),
                                      segment_len=int(rng.integers(1, 6)))
        slice_kw = {}
        if mode == "slices":  # XCD-affine column slices forced on: rows longer than 1..40 entries cut at 8..32 column boundaries
            slice_kw = dict(slice_threshold=int(rng.choice([1, 2, 5, 16, 40])), n_slices=int(rng.choice([8, 16, 24, 32])))
            seg = int(rng.choice([0, 0, 3, 16]))
            g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, segment_len=seg, split_threshold=2 * seg, **slice_kw)
        strided = mode != "plan_free" and rng.random() < 0.3  # X and Z as column slices of wider matrices
        fused = dtype == torch.float32 and mode != "plan_free" and not strided and rng.random() < 0.35
        in_launch = int(rng.choice([0, 1, 2, 2])) if fused else 0
        H_pick = None
        if in_launch == 2 and rng.random() < 0.8:  # mostly shapes the row-tile form serves (others fall back to two launches)
            D, H_pick = int(rng.choice([32, 48, 64, 96, 128])), int(rng.choice([16, 32, 64, 22, 22, 7, 50, 31, 2]))  # (widths between the tile sizes: zero-padded)
        X = rng.standard_normal((N, D)).astype(np.float32)
        if in_launch:  # the same classification, plan flagged so that dense windows (1) / every tile (2: row-tile form; one column
            # pass forced half of the time so that wide embeddings take it too) are multiplied inside the aggregation launches
            g.row_nzr = fe.build_plan(g.rp_d, g.col_d, g.bp, g.e2c, g.ht, fuse_in_launch=in_launch,
                                      panel_cols=-1 if (in_launch == 2 and rng.random() < 0.5) else 0, **slice_kw)
        tag = "%s/%s%s%s/%s" % (mode, str(dtype).replace("torch.", ""), "/strided" if strided else "",
                                ("/fused_form%d" % in_launch) if fused else "", fe.name)

        def run(Xd):
            if not strided:
                return g.forward(Xd)
            offx, offz = int(rng.choice([0, 1, 3, 4, 5, 8, 32])), int(rng.choice([0, 1, 3, 4, 5, 8, 32]))
            wide_x = torch.zeros(N, D + offx + int(rng.integers(0, 9)), dtype=Xd.dtype, device=dev)
            wide_z = torch.full((N, D + offz + int(rng.integers(0, 9))), 7.0, dtype=Xd.dtype, device=dev)
            wide_x[:, offx:offx + D] = Xd
            fe.forward_into(wide_x[:, offx:offx + D], wide_z[:, offz:offz + D], *g.args())
            assert bool((wide_z[:, :offz] == 7).all()) and bool((wide_z[:, offz + D:] == 7).all()), "wrote outside its slice"
            return wide_z[:, offz:offz + D].contiguous()
        try:
            if fused:
                H = H_pick if H_pick else int(rng.choice([16, 32, 32, 32, 7, 64, 48, 22, 50]))
                Wm = rng.standard_normal((D, H)).astype(np.float32)
                Xd, Wd = torch.from_numpy(X).to(dev), torch.from_numpy(Wm).to(dev)
                out, out2 = fe.forward_fixed32_fused(Xd, *g.args(), Wd)
                T._check(oracle, g, X, out2)
                assert torch.equal(out2, g.forward(Xd)), "fused out2 differs from the plain forward"
                want, _ = oracle.spmm_fused_f32(rp, col, X, Wm)
                scale = oracle.spmm_f64(rp, col, X, absolute=True) @ np.abs(Wm).astype(np.float64)
                assert np.all(np.abs(out.cpu().numpy().astype(np.float64) - want) <= 1e-5 * scale + 1e-30), "fused out off"
            elif dtype == torch.float32:
                T._check(oracle, g, X, run(torch.from_numpy(X).to(dev)))
            else:
                X16 = torch.from_numpy(X).to(dtype).to(dev)
                T._check_h16(oracle, g, X16, run(X16))
        except Exception:
            print("FAILED case %d: N=%d E=%d D=%d %s" % (case, N, len(col), D, tag))
            raise
        stats[tag] = stats.get(tag, 0) + 1
        if (case + 1) % 500 == 0:
            print("... %d cases ok" % (case + 1), flush=True)  # a long silent run is taken for a hang by gpurun
    torch.cuda.synchronize()
    print("fuzz ok: %d cases (seed %d)" % (args.cases, args.seed))
    for k in sorted(stats):
        print("  %-28s %d" % (k, stats[k]))


if __name__ == "__main__":
    main()
