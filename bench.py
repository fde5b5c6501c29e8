#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hybrid SpMM hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload reddit|cora|dense] [--dim D]

A "step" is one pass of the hot path, Z = A*X, over one synthetic batch (graph + embedding matrix)
already resident in HBM.  N = 1 runs BASELINE.json config "Reddit-scale": 233 000 nodes /
11.6 M stored entries, power-law, dim 128 (the configuration the metric "dim=128" is quoted on that
fits one GPU).  N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL) is weak scaling:
every rank owns one such row block (columns span all N*233 000 vertices) and each step all-gathers
the embedding row blocks over xGMI before its local product -- the one exchange step the path has.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, from
HIP-event timing of the SpMM launches on the launch stream) and, at N = 1, `cpu_baseline` (the
plain-C oracle port on one host core over a bounded row sample; torch.sparse.mm on all host cores
is reported beside it as BASELINE.json asks).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "hc-spmm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable streaming)

WORKLOADS = {
    # name: (nodes per rank, stored entries per rank, default dim, description)
    "reddit": (233000, 11600000, 128, "synthetic power-law, Reddit-scale (BASELINE config 3)"),
    "cora": (10000, 50000, 32, "synthetic power-law, Cora-scale (BASELINE config 2)"),
    "alldense": (1000000, 0, 128, "every window planted (16 rows sharing 20 columns): dense-tile path only, MFMA-utilisation probe"),
    "rd_like": (4859280, 10149830, 32, "synthetic power-law with the paper's RD size (Table II: 4.86 M nodes / 10.1 M entries), low degree"),
    "tt_like": (3771081, 22011034, 32, "synthetic power-law with the paper's TT size (3.77 M nodes / 22.0 M entries)"),
    "dp_like": (18268981, 172183984, 32, "synthetic power-law with the paper's DP size (18.3 M nodes / 172 M entries)"),
    "yh_like": (3139988, 6280000, 32, "molecule-collection graph of the paper's YeastH order (3.1 M nodes, avg degree 2, neighbours within a few dozen ids)"),
    "products_share": (306250, 7750000, 256, "one GPU's row block of BASELINE config 4 (ogbn-products scale: 2.45 M nodes / 62 M entries over 8 GPUs); use with --virtual-world 8"),
    "powerlaw16m_share": (2000000, 32000000, 128, "one GPU's row block of BASELINE config 5 (16 M nodes / 256 M entries over 8 GPUs); use with --virtual-world 8"),
    "dense": (2000000, 0, 128, "planted 16-row groups sharing <=24 columns, dense-tile heavy (BASELINE config 5 shape, per-GPU share)"),
}


def make_local_block(workload, n_local, e_local, world, rank, seed=3):
    """Row block of `rank`: n_local rows, columns are global ids in [0, world*n_local)."""
    from hcspmm import graphs
    if world == 1:
        if workload in ("dense", "alldense"):
            return graphs.planted_dense_graph_fast(n_local, seed=seed, dense_fraction=0.7 if workload == "dense" else 1.0,
                                                   k_cols=20, fill=0.45, sparse_degree=16)
        if workload == "yh_like":
            return graphs.molecule_graph(n_local, seed=seed)
        return graphs.powerlaw_graph(n_local, e_local, seed=seed)
    # rows follow a local power law, columns a global one (cheap to generate per rank, no exchange)
    rng = np.random.default_rng(seed + 1000 * rank)
    n_total = n_local * world
    alpha = 1.0 / 1.1

    def weights(n, cap_deg, e):
        w = (np.arange(n, dtype=np.float64) + 1.0) ** (-alpha)
        w /= w.sum()
        for _ in range(8):
            w = np.minimum(w, cap_deg / e)
            w /= w.sum()
        c = np.cumsum(w)
        c[-1] = 1.0
        return c
    rcdf = weights(n_local, 0.02 * n_local, e_local)
    ccdf = weights(n_total, 0.02 * n_local, e_local)
    rperm = rng.permutation(n_local)
    cperm = np.random.default_rng(seed).permutation(n_total)  # same column relabelling on every rank
    draw = int(e_local * 1.12)
    r = rperm[np.searchsorted(rcdf, rng.random(draw))].astype(np.int64)
    c = cperm[np.searchsorted(ccdf, rng.random(draw))].astype(np.int64)
    key = np.unique(r * n_total + c)
    if key.shape[0] > e_local:
        key = np.sort(rng.choice(key, e_local, replace=False))
    rows, cols = key // n_total, key % n_total
    rp = np.zeros(n_local + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp).astype(np.int32), cols.astype(np.int32)


def algorithmic_bytes(N, E, D, header, elem=4):
    """4*E*D row gathers + 4*N*D Z write + 4*E column ids + 4*(N+1) row pointers; for
    dense-path windows 4*uniq_w*D instead of 4*nnz_w*D, plus 8*nnz_w (edgeToColumn + edgeToRow).
    (elem = bytes per feature element: 4, or 2 with --dtype f16 / bf16.)"""
    nnz_d, uniq_d = header.nnz_dense, header.uniq_dense
    return float(elem) * ((E - nnz_d) * D + uniq_d * D + N * D) + 8.0 * nnz_d + 4.0 * E + 4.0 * (N + 1)


def cpu_baseline(rp, col, X_host, D, budget_s=12.0):
    """Oracle port (plain C, one core) on a bounded prefix of the rows + torch.sparse.mm (all cores)."""
    import oracle
    N = len(rp) - 1
    rows = min(N, 4096)
    out = {}
    t_used, edges, reps = 0.0, 0, 0
    while True:  # grow the sample until ~1/3 of the budget is one call, then repeat
        t0 = time.perf_counter()
        oracle.spmm_f32(rp[:rows + 1], col[:rp[rows]], X_host)
        dt = time.perf_counter() - t0
        if dt > budget_s / 4 or rows == N:
            t_used, edges, reps = dt, int(rp[rows]), 1
            break
        rows = min(N, rows * 4)
    while t_used < budget_s / 2 and reps < 5:
        t0 = time.perf_counter()
        oracle.spmm_f32(rp[:rows + 1], col[:rp[rows]], X_host)
        t_used += time.perf_counter() - t0
        reps += 1
    out = {"value": edges * reps * D / t_used, "unit": "edge*dim/s", "cores": 1, "kind": "port",
           "sample": "oracle/hcspmm_oracle.c spmm_f32 on the first %d of %d rows (%d entries), %d reps" % (rows, N, edges, reps)}
    try:
        A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                    torch.ones(len(col)), size=(N, X_host.shape[0]))
        Xt = torch.from_numpy(X_host)
        torch.sparse.mm(A, Xt)
        t0 = time.perf_counter()
        n = 0
        while n < 3 and time.perf_counter() - t0 < budget_s / 2:
            torch.sparse.mm(A, Xt)
            n += 1
        dt = (time.perf_counter() - t0) / max(n, 1)
        out["torch_sparse_mm"] = {"value": len(col) * D / dt, "unit": "edge*dim/s", "threads": torch.get_num_threads(),
                                  "host_cores": os.cpu_count(), "ms": dt * 1e3}
    except Exception as e:  # reported, never fatal
        out["torch_sparse_mm"] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # SAG.profile's round count, GNN_model.py:251
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="reddit", choices=sorted(WORKLOADS))
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"],
                    help="feature element type; f32 is the reference's (and BASELINE's) -- the 16-bit variants are the paper's Table VII extension")
    ap.add_argument("--rule", type=int, default=0, help="window classifier (hcspmm.h: 0 intended, 2 as shipped = all sparse, 3 MI355X refit)")
    ap.add_argument("--no-plan", action="store_true", help="use the plan-free (reference-convention) kernel")
    ap.add_argument("--virtual-world", type=int, default=1,
                    help="one-GPU run of ONE rank's local product in a P-GPU job: the row block references columns of "
                         "all P blocks and X holds all P*n rows (already 'gathered'); no communication is timed")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
              % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # HCSPMM_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo collectives -- lets the N > 1 plumbing run
    # on a one-GPU box (numbers from it are meaningless and are labelled as such)
    rehearsal = os.environ.get("HCSPMM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import hcspmm
    from hcspmm.sharded import ShardedGraph, ShardedSpMM

    n_local, e_local, d_default, desc = WORKLOADS[args.workload]
    D = args.dim or d_default
    vworld = max(1, args.virtual_world) if world == 1 else 1
    rp, col = make_local_block(args.workload, n_local, e_local, world * vworld, rank)
    g = ShardedGraph.from_local_block(rp, col, n_local, world, rank)
    E = int(len(col))
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    t0 = time.perf_counter()
    bp, e2c, e2r, ht, row_nzr, col_nzr = hcspmm.preprocess(col_d, rp_d, n_local, E, (n_local + 15) // 16, rule=args.rule,
                                                           num_columns=n_local * world * vworld)
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3
    header = hcspmm.plan_header(row_nzr)
    if args.no_plan:
        row_nzr = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.manual_seed(1234 + rank)
    X_local = torch.randn(n_local * vworld, D, device=dev)  # dataset.py:114 init_embedding
    tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    X_local = X_local.to(tdtype)
    elem = X_local.element_size()

    ev_pairs = []

    def timed(fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()  # torch's current stream == the stream hcspmm launches on
        out = fn()
        e.record()
        ev_pairs.append((s, e))
        return out

    def local_spmm(X_full):
        return timed(lambda: hcspmm.forward_rect(X_full, rp_d, col_d, bp, e2c, e2r, ht, row_nzr, col_nzr)[0])

    def local_spmm_into(X_panel_full, Z_view):
        return timed(lambda: hcspmm.forward_into(X_panel_full, Z_view, rp_d, col_d, bp, e2c, e2r, ht, row_nzr, col_nzr))

    # N > 1: gather X in panels of one cache line per row (32 fp32 / 64 16-bit columns) and multiply panel k under
    # the gather of panel k+1
    n_gather_panels = int(os.environ.get("HCSPMM_GATHER_PANELS", "0"))
    line_cols = 128 // elem
    if n_gather_panels <= 0:
        n_gather_panels = D // line_cols if (world > 1 and D >= 2 * line_cols and D % line_cols == 0) else 1
    op = ShardedSpMM(g, local_spmm, local_spmm_into=local_spmm_into, n_panels=n_gather_panels)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        op(X_local)
    sync_all()
    ev_pairs.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Z = op(X_local)
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.sum([s.elapsed_time(e) for s, e in ev_pairs])) / args.steps if ev_pairs else float("nan")

    red_dev = torch.device("cpu") if rehearsal else dev
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    etot = torch.tensor([float(E)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(etot, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_edges = float(etot.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        b_alg = algorithmic_bytes(n_local, E, D, header, elem)
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1 and args.dtype == "f32" and args.rule == 0:  # measured for the default one-GPU launch only
            try:
                traffic = json.load(open(tpath)).get("%s_d%d" % (args.workload, D))
            except Exception:
                traffic = None
        out = {
            "metric": "GNN-aggregation SpMM edges*dim/s (A*X, %s)" % ("fp32" if args.dtype == "f32" else args.dtype + " features, fp32 accumulation"),
            "value": total_edges * D / (elapsed / args.steps),
            "unit": "edge*dim/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)" if rehearsal else ""),
            "config": {"workload": "%s: %s; %d nodes / %d stored entries per GPU, dim %d%s"
                                   % (args.workload, desc, n_local, E, D,
                                      "" if vworld == 1 else "; ONE rank of a virtual %d-GPU job (X: %d rows resident)" % (vworld, n_local * vworld)),
                       "nodes_per_gpu": n_local, "entries_per_gpu": E, "dim": D, "parallelism": "row-block shard x%d + all-gather(X) in %d column panel(s)" % (world, n_gather_panels if world > 1 else 1),
                       "plan": (not args.no_plan), "rule": args.rule, "sparse_tasks": header.n_tasks, "dense_windows": header.n_dense,
                       "split_rows": header.n_split_rows, "preprocess_ms": prep_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "hcspmm::hybrid_plan_kernel (+ fixup_kernel)", "kernel_ms": kern_ms,
                         "algorithmic_bytes": b_alg,
                         "traffic_gbs": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                         "note": "achieved counts every gathered X row (SURVEY 8d); frac > 1 means rows were served by "
                                 "L2 / Infinity Cache instead of HBM -- `traffic` is what actually crossed the fabric "
                                 "(PMC, profiles/), and the launch is bound by that"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rp, col, X_local.float().cpu().numpy(), D)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
