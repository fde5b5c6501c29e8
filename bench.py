#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hybrid SpMM hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--dim D] [--frontend ctypes|extension]

A "step" is one pass of the hot path, Z = A*X, over one synthetic batch (graph + embedding matrix)
already resident in HBM.  N = 1 runs BASELINE.json config 3, "Reddit-scale": 233 000 nodes /
11.6 M stored entries, power-law, dim 128 (the configuration the metric "dim=128" is quoted on that
fits one GPU).  N > 1 -- one rank per GPU over RCCL; `python bench.py --gpus N` starts its N ranks itself
(fresh child processes, before anything touches a GPU), `python -m torch.distributed.run ... bench.py --gpus N`
works as well -- is weak scaling: every rank owns one such row block (columns span all N*233 000 vertices)
and each step all-gathers the embedding row blocks over xGMI before its local product -- the one exchange
step the path has.  `--workload products | powerlaw16m | c5` are the STRONG-scaling spellings of BASELINE
configs 4 and 5: the same full graph (2.45 M / 62 M, 16 M / 256 M) at every N, its rows split over the ranks
in nnz-balanced, window-aligned blocks.

Prints ONE JSON line on rank 0 (contract in the task description).  At N = 1 the same line carries
  * `roofline`: live HIP-event kernel time, and -- measured by THIS run -- the bytes that crossed the fabric
    per launch: before the process touches the GPU it runs itself under `rocprofv3 --pmc` (FETCH_SIZE,
    WRITE_SIZE, TCC hit/miss: separate passes, as MI355X_MICROARCH.md prescribes) on the same graph and reads
    the counters back.  `frac` is that traffic / time / 8 TB/s (<= 1 by construction); the algorithmic and
    compulsory figures stand beside it.  If the profiler is unavailable the recorded figure under profiles/
    is used and labelled with its source; failing that, the compulsory-bytes fraction.
  * `sweep`: the other BASELINE points timed in the same process -- Reddit-scale at dim 32 and 256, the
    Cora-scale config 2, one GPU's share of config 4 (dim 256) and one GPU's share of config 5 as SURVEY.md 8(d)
    defines it (planted <= 24-column groups over 16 M columns: most windows on the dense-tile / MFMA path), plus an
    RD-sized low-degree graph (the paper's Table II) and the all-dense MFMA-utilisation probe -- each
    with ITS counters from the same child passes (one rocprofv3 run per counter set covers every workload);
  * `cpu_baseline`: torch.sparse.mm on the box's host cores (threads stated), the oracle port nested beside it.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "hc-spmm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md, L2 (per XCD): 4 MiB x 8, ~34.5 TB/s aggregate
HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable streaming)
FP32_MFMA_PEAK_TFLOPS = 157.3
INFINITY_CACHE_BYTES = 256 << 20

WORKLOADS = {
    # name: (nodes per rank, stored entries per rank, default dim, virtual world, description)
    "reddit": (233000, 11600000, 128, 1, "synthetic power-law, Reddit-scale (BASELINE config 3)"),
    "cora": (10000, 50000, 32, 1, "synthetic power-law, Cora-scale (BASELINE config 2)"),
    "alldense": (1000000, 0, 128, 1, "every window planted (16 rows sharing 20 columns): dense-tile path only, MFMA-utilisation probe"),
    "rd_like": (4859280, 10149830, 32, 1, "synthetic power-law with the paper's RD size (Table II: 4.86 M nodes / 10.1 M entries), low degree"),
    "tt_like": (3771081, 22011034, 32, 1, "synthetic power-law with the paper's TT size (3.77 M nodes / 22.0 M entries)"),
    "dp_like": (18268981, 172183984, 32, 1, "synthetic power-law with the paper's DP size (18.3 M nodes / 172 M entries)"),
    "yh_like": (3139988, 6280000, 32, 1, "molecule-collection graph of the paper's YeastH order (3.1 M nodes, avg degree 2, neighbours within a few dozen ids)"),
    "products_share": (306250, 7750000, 256, 8, "one GPU's row block of BASELINE config 4 (ogbn-products scale: 2.45 M nodes / 62 M entries over 8 GPUs; all 2.45 M X rows resident)"),
    "powerlaw16m_share": (2000000, 32000000, 128, 8, "one GPU's row block of a 16 M-node / 256 M-entry power-law graph WITHOUT planted groups (all sparse-row)"),
    "c5_share": (2000000, 32000000, 128, 8, "one GPU's row block of BASELINE config 5 (16 M nodes / 256 M entries over 8 GPUs; all 16 M X rows resident): "
                                           "70 % of the 16-row windows are planted groups sharing 8-24 columns (dense-tile path under the reference's classifier), the rest power-law rows"),
    "community": (4859280, 10149830, 128, 1, "community-structured graph of the paper's RD size (4.86 M nodes / ~10 M entries): planted groups of 8-40 rows "
                                              "sharing a column pool of their own members + power-law noise, vertex ids SHUFFLED (hcspmm.graphs.community_graph)"),
    "community_loi": (4859280, 10149830, 128, 1, "the same graph after hcspmm.loi_reorder(variant='fast') + apply_permutation (the LOI layout reorder, LOI.cpp:660-805, relaxed parallel form)"),
    "dense": (2000000, 0, 128, 1, "2 M-node square graph, 70 % planted windows of 20 columns + 16 random entries per other row (round-1 dense-heavy proxy)"),
}

# Strong-scaling spellings of BASELINE configs 4 and 5: name -> (total nodes, total stored entries, dim, description).
# The graph is the concatenation of STRONG_CHUNKS row chunks of equal height and equal entry count, chunk c generated from
# (seed, c) alone: every world size that divides STRONG_CHUNKS sees the SAME graph, and the contiguous chunk ranges are
# the nnz-balanced, window-aligned blocks hcspmm.sharded.partition_rows cuts it into -- exactly for the power-law graphs,
# to within a row window or two for the planted one (tests/test_bench_cpu.py).
STRONG_CHUNKS = 64
STRONG_WORKLOADS = {
    "products": (38288 * 64, 968750 * 64, 256, "BASELINE config 4 at full size (ogbn-products scale: 2.45 M nodes / 62 M entries, power-law), rows split over the ranks"),
    "powerlaw16m": (250000 * 64, 4000000 * 64, 128, "16 M-node / 256 M-entry power-law graph without planted groups (all sparse-row), rows split over the ranks"),
    "c5": (250000 * 64, 4000000 * 64, 128, "BASELINE config 5 at full size (16 M nodes / 256 M entries; 70 % of the 16-row windows planted groups sharing 8-24 "
                                           "columns: dense-tile path under the reference's classifier), rows split over the ranks"),
}
SWEEP_PLAN = [("reddit", 32), ("reddit", 256), ("cora", 32), ("products_share", 256), ("c5_share", 128),
              ("rd_like", 32), ("alldense", 128)]  # + one of the paper's Table-II shapes (low degree) and the dense-tile / MFMA probe

KERNEL_SOURCES = ["hc-spmm_amd/csrc/spmm_impl.h", "hc-spmm_amd/csrc/spmm_kernels.h", "hc-spmm_amd/csrc/capi.hip",
                  "hc-spmm_amd/csrc/plan_host.cpp", "hc-spmm_amd/csrc/preprocess_host.cpp", "include/hcspmm.h"]


def kernel_src_sha():
    """Identifies the kernel + plan sources a recorded profile belongs to (profiles/measured.json entries carry it)."""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def make_local_block(workload, n_local, e_local, world, rank, seed=3):
    """Row block of `rank`: n_local rows, columns are global ids in [0, world*n_local)."""
    from hcspmm import graphs
    if workload == "c5_share":
        return graphs.planted_powerlaw_block(n_local, n_local * world, e_local, seed=seed, rank=rank)
    if world == 1:
        if workload in ("dense", "alldense"):
            return graphs.planted_dense_graph_fast(n_local, seed=seed, dense_fraction=0.7 if workload == "dense" else 1.0,
                                                   k_cols=20, fill=0.45, sparse_degree=16)
        if workload == "yh_like":
            return graphs.molecule_graph(n_local, seed=seed)
        if workload == "community":
            return graphs.community_graph(n_local, e_local, seed=seed)[:2]
        return graphs.powerlaw_graph(n_local, e_local, seed=seed)
    return graphs.powerlaw_block(n_local, n_local * world, e_local, seed=seed, rank=rank)


def _strong_chunk(job):
    """One row chunk of a strong-scaling graph (a top-level function: worker processes of make_strong_block import it)."""
    workload, n_chunk, n_total, e_chunk, seed, c = job
    from hcspmm import graphs
    if workload == "c5":
        return graphs.planted_powerlaw_block(n_chunk, n_total, e_chunk, seed=seed, rank=c)
    return graphs.powerlaw_block(n_chunk, n_total, e_chunk, seed=seed, rank=c)


def make_strong_block(workload, world, rank, seed=3, chunks=STRONG_CHUNKS, scale=1, workers=1):
    """Rows [rank, rank + 1) * N / world of the strong-scaling graph `workload` (columns global): the concatenation of this
    rank's chunks.  scale > 1 shrinks the graph (tests).  workers > 1: the chunks -- each a function of (seed, chunk id) alone --
    are generated by that many fresh host processes (spawned, so they never inherit a GPU context).
    -> (row_pointers, column_index, n_local, n_total)."""
    n_total, e_total, _, _ = STRONG_WORKLOADS[workload]
    if chunks % world != 0:
        raise SystemExit("bench.py: --workload %s splits %d row chunks: --gpus must divide that (1, 2, 4, 8, ...)" % (workload, chunks))
    n_chunk, e_chunk = n_total // chunks // scale // 16 * 16, e_total // chunks // scale
    n_total = n_chunk * chunks
    per = chunks // world
    jobs = [(workload, n_chunk, n_total, e_chunk, seed, c) for c in range(rank * per, (rank + 1) * per)]
    rps, cols, base = [np.zeros(1, np.int64)], [], 0
    if workers > 1:
        # plain child interpreters (not multiprocessing: nothing of this process -- its __main__, its GPU context -- is inherited),
        # each generating every workers-th chunk into a scratch directory
        tmp = tempfile.mkdtemp(prefix="hcspmm_chunks_")
        code = ("import sys, json, numpy as np; sys.path[:0] = %r; import bench\n"
                "for i, job in enumerate(json.loads(sys.argv[1])):\n"
                "    if i %% int(sys.argv[3]) == int(sys.argv[2]):\n"
                "        rp, col = bench._strong_chunk(tuple(job)); np.save('%s/rp_%%d.npy' %% i, rp); np.save('%s/col_%%d.npy' %% i, col)\n"
                % ([ROOT, os.path.join(ROOT, "hc-spmm_amd")], tmp, tmp))
        n_proc = min(workers, len(jobs))
        procs = [subprocess.Popen([sys.executable, "-c", code, json.dumps(jobs), str(w), str(n_proc)]) for w in range(n_proc)]
        rcs = [p.wait() for p in procs]
        if any(rcs):
            shutil.rmtree(tmp, ignore_errors=True)
            raise RuntimeError("bench.py: a chunk generator exited with %s" % rcs)
        parts = [(np.load(os.path.join(tmp, "rp_%d.npy" % i)), np.load(os.path.join(tmp, "col_%d.npy" % i))) for i in range(len(jobs))]
        shutil.rmtree(tmp, ignore_errors=True)
    else:
        parts, t_last = [], time.time()
        for i, job in enumerate(jobs):
            if rank == 0 and time.time() - t_last > 30.0:  # the 16 M-node graph takes minutes on one rank: say that the run is alive
                print("bench.py: generating %s, row chunk %d of %d" % (workload, i, per), file=sys.stderr, flush=True)
                t_last = time.time()
            parts.append(_strong_chunk(job))
    for rp, col in parts:
        rps.append(rp[1:].astype(np.int64) + base)
        cols.append(col)
        base += int(rp[-1])
    return np.concatenate(rps).astype(np.int32), np.concatenate(cols).astype(np.int32), n_chunk * per, n_total


def algorithmic_bytes(N, E, D, header, elem=4):
    """4*E*D row gathers + 4*N*D Z write + 4*E column ids + 4*(N+1) row pointers; for
    dense-path windows 4*uniq_w*D instead of 4*nnz_w*D, plus 8*nnz_w (edgeToColumn + edgeToRow).
    (elem = bytes per feature element: 4, or 2 with --dtype f16 / bf16.)"""
    nnz_d, uniq_d = header.nnz_dense, header.uniq_dense
    return float(elem) * ((E - nnz_d) * D + uniq_d * D + N * D) + 8.0 * nnz_d + 4.0 * E + 4.0 * (N + 1)


def compulsory_bytes(N, E, D, n_cols_referenced, elem=4):
    """Every referenced X row read once + Z written once + the CSR arrays once (SURVEY.md 8(d) B_min)."""
    return float(elem) * (n_cols_referenced + N) * D + 4.0 * E + 4.0 * (N + 1)


# ------------------------------------------------------------------------------------------------
# front-ends over the C ABI: the ctypes glue (default) or the reference's own boundary, the HCSPMM extension
# ------------------------------------------------------------------------------------------------
class _Frontend:
    def __init__(self, name):
        self.name = name
        if name == "extension":
            ext = os.path.join(ROOT, "hc-spmm_amd", "hybrid_kernel")
            if ext not in sys.path:
                sys.path.insert(0, ext)
            import HCSPMM
            self.m = HCSPMM
        else:
            import hcspmm
            self.m = hcspmm

    def preprocess(self, col_d, rp_d, N, E, W, rule, num_columns):
        if self.name == "extension":
            before = self.m.get_rule()
            self.m.set_rule(int(rule))
            try:
                return self.m.preprocess(col_d, rp_d, N, E, W, int(num_columns))
            finally:
                self.m.set_rule(before)
        return self.m.preprocess(col_d, rp_d, N, E, W, rule=rule, num_columns=num_columns)

    def header(self, row_nzr):
        if self.name == "extension":
            return types.SimpleNamespace(**self.m.plan_info(row_nzr))
        return self.m.plan_header(row_nzr)

    def workspace_bytes(self, row_nzr, w):
        if self.name == "extension":
            return 4 * int(self.m.plan_info(row_nzr).get("n_partials", 0)) * int(w)
        return self.m.workspace_bytes(row_nzr, w)

    def forward_into(self, X, Z, graph_args, workspace):
        if self.name == "extension":
            return self.m.forward_into(X, Z, *graph_args, workspace)
        return self.m.forward_into(X, Z, *graph_args, workspace=workspace)


# ------------------------------------------------------------------------------------------------
# rocprofv3 counter passes of this very benchmark, run as child processes BEFORE this process touches the GPU
# ------------------------------------------------------------------------------------------------
PMC_PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "l2": ["TCC_HIT_sum", "TCC_MISS_sum"],
              "mfma": ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]}


def case_key(workload, D, dtype="f32", rule=0):
    return "%s_d%d%s%s" % (workload, D, "" if dtype == "f32" else "_" + dtype, "" if not rule else "_r%d" % rule)


DEFAULT_RULE = 3  # the library's default classifier: the width-agnostic MI355X refit (hcspmm.default_rule())


def mi355x_rule(D):
    """hcspmm.mi355x_rule without importing the package before the counter passes: 3 = the narrow refit (D < 64), 4 = the wide one."""
    return 3 if int(D) < 64 else 4


def loi_plan():
    """The `loi` block's cases: {as shuffled, after the reorder} x {the reference's classifier, the MI355X refit} at the headline
    width and at the paper's (GNN_model.py:39 calls the D = 32 kernel)."""
    plan = [(w, D, r) for D in (128, 32) for w in ("community", "community_loi") for r in (0, mi355x_rule(D))]
    # ... and the reordered graph with every window forced onto the sparse-row path (rule 2, the reference's as-shipped line): what
    # the dense-tile / MFMA path itself contributes on top of the locality the reorder brings
    return plan + [("community_loi", D, 2) for D in (128, 32)]


def _read_counter_segments(d):
    """Counters of one rocprofv3 pass, cut into one segment per benchmark case.  The child runs its cases one after the
    other and every case begins with exactly one preprocess, whose first launch is the library's edge_to_row_kernel: that
    dispatch marks the start of a segment.  -> [ {counter: {kernel: mean value per launch}} ] in case order."""
    import collections
    import csv
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Dispatch_Id"]), r.get("Kernel_Name", ""), r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort(key=lambda r: r[0])
    segs, marker_ids = [], set()
    for did, name, counter, value in rows:
        if "edge_to_row_kernel" in name:
            if did not in marker_ids:
                marker_ids.add(did)
                segs.append(collections.defaultdict(lambda: collections.defaultdict(list)))
        elif "hcspmm::" in name and segs:
            segs[-1][counter][name.split("(")[0]].append(value)
    return [{c: {k: sum(v) / len(v) for k, v in per.items()} for c, per in seg.items()} for seg in segs]


def live_pmc(args, cache_dir, cases, passes):
    """-> {case key: {"fetch_bytes", "write_bytes", "traffic_bytes", "l2_hit_rate", "mfma_util_percent", ...}} plus
    "_passes" (seconds per pass), or {"error": ...}.  Each pass is ONE run of
    `rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --pmc-child --cases ...` (the program directly after `--`)
    that executes every case in turn on the graphs this process has already generated (cache_dir)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    # already inside a profiler (this very command run under rocprofv3, e.g. by tools/profile_round.sh or a harness)?  Then no
    # nested profiler runs: the recorded figures are used instead
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "already running under a profiler: live counter passes skipped"}
    out = {"_passes": {}}
    child = ["python3", os.path.join(ROOT, "bench.py"), "--pmc-child", "--graph-cache", cache_dir,
             "--cases", ",".join("%s:%d:%d" % (c[0], c[1], c[2] if len(c) > 2 else 0) for c in cases), "--workload", args.workload, "--dim", str(args.dim),
             "--steps", "4", "--warmup", "2", "--rule", str(args.rule), "--dtype", args.dtype,
             "--frontend", args.frontend, "--virtual-world", str(args.virtual_world)] + (["--no-plan"] if args.no_plan else [])
    env = dict(os.environ, TMPDIR="/tmp")
    keys = [case_key(c[0], c[1], args.dtype if i == 0 else "f32", (c[2] if len(c) > 2 else 0) if i else args.rule) for i, c in enumerate(cases)]
    keys[0] = case_key(cases[0][0], cases[0][1], args.dtype)  # (the headline is looked up without its rule suffix)
    per_case = {k: {"counters": {}} for k in keys}
    budget = float(os.environ.get("HCSPMM_BENCH_PMC_TIMEOUT", "240"))
    for name in passes:
        d = os.path.join(cache_dir, "pmc_" + name)
        cmd = [rocprof, "--pmc"] + PMC_PASSES[name] + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
        t0 = time.perf_counter()
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=budget)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.wait()
                return {"error": "rocprofv3 pass '%s' timed out" % name}
            if p.returncode != 0:
                return {"error": "rocprofv3 pass '%s' exited %d: %s" % (name, p.returncode, log.decode(errors="replace")[-300:])}
        except Exception as e:  # profiler trouble is reported, never fatal
            return {"error": "rocprofv3 pass '%s': %s" % (name, str(e)[:200])}
        segs = _read_counter_segments(d)
        if len(segs) != len(cases) or not all(segs):
            return {"error": "rocprofv3 pass '%s': %d counter segments for %d cases" % (name, len(segs), len(cases))}
        out["_passes"][name] = round(time.perf_counter() - t0, 1)
        for k, seg in zip(keys, segs):
            per_case[k]["counters"].update(seg)
        shutil.rmtree(d, ignore_errors=True)
    for k in keys:
        c = per_case[k]["counters"]
        e = {}
        # MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
        # of wide (16 B / lane) reads -> doubled; WRITE_SIZE is exact for 16-byte stores.  Summed over the launches of a
        # step (hybrid kernel + fix-up); tools/calibrate_fetch_size.py checks the factor on a copy of known size.
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["fetch_bytes"] = 2.0 * 1024.0 * sum(c["FETCH_SIZE"].values())
            e["write_bytes"] = 1024.0 * sum(c["WRITE_SIZE"].values())
            e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            hit, miss = sum(c["TCC_HIT_sum"].values()), sum(c["TCC_MISS_sum"].values())
            e["l2_hit_rate"] = hit / max(hit + miss, 1.0)
            e["l2_request_bytes"] = 128.0 * (hit + miss)  # 128-byte lines through the eight L2s per step
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # rocprofv3's MfmaUtil: MFMA-busy cycles summed over the SIMDs / (GPU-active cycles x SIMDs); GRBM_GUI_ACTIVE is
            # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note).  The hybrid kernel only (the fix-up has no MFMA).
            hk = [k2 for k2 in c["GRBM_GUI_ACTIVE"] if "hybrid" in k2]
            if hk and c["GRBM_GUI_ACTIVE"][hk[0]] > 0:
                e["mfma_util_percent"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"].get(hk[0], 0.0) / (c["GRBM_GUI_ACTIVE"][hk[0]] / 8.0 * 1024.0)
        out[k] = e
    return out


def recorded_profile(key):
    """Entry of profiles/measured.json (written by profiles/summarize.py from a builder-run rocprofv3 session)."""
    path = os.path.join(ROOT, "profiles", "measured.json")
    if not os.path.exists(path):
        return None
    try:
        e = json.load(open(path)).get(key)
    except Exception:
        return None
    if e:
        e = dict(e)
        e["stale"] = e.get("kernel_src_sha") != kernel_src_sha()
    return e


# ------------------------------------------------------------------------------------------------
def cpu_baseline(rp, col, X_host, D, n_cols, budget_s=20.0):
    """torch.sparse.mm (CSR, all host threads torch uses) on the full workload + the plain-C oracle port on one core
    over a bounded prefix of the rows."""
    import torch
    import oracle
    N = len(rp) - 1
    out = {"unit": "edge*dim/s", "kind": "torch.sparse.mm", "host_cores": os.cpu_count()}
    try:
        A = torch.sparse_csr_tensor(torch.from_numpy(rp.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                    torch.ones(len(col)), size=(N, n_cols))
        Xt = torch.from_numpy(X_host)
        torch.sparse.mm(A, Xt)  # warm-up
        t0 = time.perf_counter()
        n = 0
        while n < 5 and time.perf_counter() - t0 < budget_s * 0.6:
            torch.sparse.mm(A, Xt)
            n += 1
        dt = (time.perf_counter() - t0) / max(n, 1)
        out.update({"value": len(col) * D / dt, "cores": torch.get_num_threads(), "ms": dt * 1e3,
                    "sample": "torch.sparse.mm(A_csr, X) on the whole workload (%d rows, %d entries, dim %d), %d timed calls after "
                              "one warm-up, %d threads of %d host cores" % (N, len(col), D, n, torch.get_num_threads(), os.cpu_count())})
    except Exception as e:  # reported, never fatal
        out.update({"value": None, "cores": 0, "sample": "torch.sparse.mm failed: " + str(e)[:200]})
    rows = min(N, 4096)
    t_used, edges, reps = 0.0, 0, 0
    port_budget = budget_s * 0.4
    while True:  # grow the sample until one call is ~1/4 of the budget, then repeat
        t0 = time.perf_counter()
        oracle.spmm_f32(rp[:rows + 1], col[:rp[rows]], X_host)
        dt = time.perf_counter() - t0
        if dt > port_budget / 4 or rows == N:
            t_used, edges, reps = dt, int(rp[rows]), 1
            break
        rows = min(N, rows * 4)
    while t_used < port_budget / 2 and reps < 5:
        t0 = time.perf_counter()
        oracle.spmm_f32(rp[:rows + 1], col[:rp[rows]], X_host)
        t_used += time.perf_counter() - t0
        reps += 1
    out["oracle_port"] = {"value": edges * reps * D / t_used, "unit": "edge*dim/s", "cores": 1, "kind": "port",
                          "sample": "oracle/hcspmm_oracle.c spmm_f32 on the first %d of %d rows (%d entries), %d reps" % (rows, N, edges, reps)}
    return out


# ------------------------------------------------------------------------------------------------
def run_case(fe, dev, workload, D, rp, col, n_local, world, rank, vworld, steps, warmup, dtype_name="f32", rule=0,
             no_plan=False, n_gather_panels=0, dist=None, prep_runs=3, keep_op=False):
    """Preprocess + `steps` timed steps of one workload on this rank; returns the measurements (no printing)."""
    import torch
    from hcspmm.sharded import ShardedGraph, ShardedSpMM
    E = int(len(col))
    n_cols = n_local * world * vworld
    g = ShardedGraph.from_local_block(rp, col, n_local, world, rank)
    if world == 1:
        # one GPU: X holds every row the block references (vworld > 1: the other blocks' rows are "already gathered")
        g.pad_rows = n_cols
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    W = (n_local + 15) // 16
    prep = []
    for _ in range(prep_runs):  # cold (first touch: pinned staging buffers, code objects, allocator growth), then warm (the faster of two:
        # the host passes are a few milliseconds on up to 64 threads of a shared host, and one descheduled thread doubles a single sample)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bp, e2c, e2r, ht, row_nzr, col_nzr = fe.preprocess(col_d, rp_d, n_local, E, W, rule, n_cols)
        torch.cuda.synchronize()
        prep.append((time.perf_counter() - t0) * 1e3)
    header = fe.header(row_nzr)
    plan_for_ws = row_nzr
    if no_plan:
        row_nzr = torch.zeros(1, dtype=torch.int32, device=dev)
    graph_args = (rp_d, col_d, bp, e2c, e2r, ht, row_nzr, col_nzr)
    tdtype = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype_name]
    elem = torch.empty(0, dtype=tdtype).element_size()
    ev_pairs = []

    def local_spmm_into(X_panel_full, Z_view, workspace):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()  # torch's current stream == the stream hcspmm launches on
        fe.forward_into(X_panel_full, Z_view, graph_args, workspace)
        e.record()
        ev_pairs.append((s, e))

    # N > 1: gather X in panels of one cache line per row (32 fp32 / 64 16-bit columns) and multiply panel k under
    # the gather of panel k+1
    line_cols = 128 // elem
    if n_gather_panels <= 0:
        n_gather_panels = D // line_cols if (world > 1 and D >= 2 * line_cols and D % line_cols == 0) else 1
    op = ShardedSpMM(g, local_spmm_into, n_panels=n_gather_panels,
                     workspace_bytes=(None if no_plan else (lambda w: fe.workspace_bytes(plan_for_ws, w))))
    op.bind(D, tdtype, dev)
    torch.manual_seed(1234 + rank)
    op.X_pm.normal_()  # dataset.py:114 init_embedding (x = randn): features are written panel-major, in place

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        op.step()
    sync_all()
    ev_pairs.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        op.step()
    sync_all()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.sum([s.elapsed_time(e) for s, e in ev_pairs])) / steps if ev_pairs else float("nan")
    graph_ms = None
    if world == 1 and E <= 2_000_000:
        # a launch-bound workload (Cora-scale: the kernel is a few microseconds, a Python call several times that): the
        # same steps captured in a HIP graph -- the operator neither synchronises nor allocates -- and replayed
        try:
            reps = 50
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(reps):
                        for p in range(op.n_panels):
                            fe.forward_into(op.X_pm[p], op.Z_pm[p], graph_args, op.workspace)
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gr.replay()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - t0) / (10 * reps) * 1e3
        except Exception:  # reported as absent, never fatal
            graph_ms = None
    flags = np.zeros(n_cols, dtype=bool)
    flags[col] = True
    kept = {"op": op} if keep_op else {}
    return {**kept, "workload": workload, "D": D, "N": n_local, "E": E, "n_cols": n_cols, "elem": elem, "header": header,
            "elapsed": elapsed, "steps": steps, "kernel_ms": kern_ms, "prep_cold_ms": prep[0], "prep_warm_ms": min(prep[1:]) if len(prep) > 1 else prep[0],
            "n_gather_panels": n_gather_panels, "cols_referenced": int(flags.sum()), "x_rows": n_cols,
            "hip_graph_ms_per_step": graph_ms}


def fold_columns(rp, col, fold=2048):
    """The same rows with every column id taken modulo `fold` (ascending inside a row, duplicates kept: same entry count,
    same task lengths).  X then spans fold lines per 32-column panel -- 256 KB -- so every gather is an L2 hit and the launch
    time that remains is the L2 -> CU gather path plus tasks, indices and stores: the floor of this degree sequence."""
    rows = np.repeat(np.arange(len(rp) - 1, dtype=np.int64), np.diff(rp))
    c2 = col.astype(np.int64) % fold
    return c2[np.lexsort((c2, rows))].astype(np.int32)


def roofline_of(case, traffic=None, traffic_source=None, l2_request_bytes=None, all_hit_ms=None):
    """The roofline block of one measured case (definitions: DESIGN.md section 5)."""
    h, D, N, E, elem = case["header"], case["D"], case["N"], case["E"], case["elem"]
    t = case["kernel_ms"] * 1e-3
    b_alg = algorithmic_bytes(N, E, D, h, elem)
    b_min = compulsory_bytes(N, E, D, case["cols_referenced"], elem)
    x_bytes = float(case["x_rows"]) * D * elem
    cache_resident = x_bytes <= INFINITY_CACHE_BYTES
    achieved = b_alg / t / 1e9
    frac_alg = achieved / HBM_PEAK_GBS
    frac_min = b_min / t / 1e9 / HBM_PEAK_GBS
    r = {"bound": "l2-miss / Infinity Cache" if cache_resident else "hbm",
         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "kernel": "hcspmm::hybrid_plan_kernel (+ fixup_kernel)", "kernel_ms": case["kernel_ms"],
         "algorithmic_bytes": b_alg, "compulsory_bytes": b_min, "x_bytes": x_bytes,
         "frac_algorithmic": frac_alg, "frac_hbm_compulsory": frac_min}
    if traffic:
        r["traffic"] = traffic
        r["traffic_gbs"] = traffic / t / 1e9
        r["traffic_source"] = traffic_source
        r["frac"] = r["traffic_gbs"] / HBM_PEAK_GBS
        r["frac_basis"] = "measured fabric (L2-miss) bytes / kernel time / peak"
        r["traffic_over_compulsory"] = traffic / b_min
    elif not cache_resident and frac_alg <= 1.0:
        r["traffic"] = None
        r["frac"] = frac_alg
        r["frac_basis"] = "algorithmic bytes / kernel time / peak (X exceeds the Infinity Cache; no counter run available)"
    else:
        r["traffic"] = None
        r["frac"] = frac_min
        r["frac_basis"] = ("compulsory bytes / kernel time / peak (no counter run available, and the algorithmic figure counts "
                           "gathers that the caches served)")
    r["note"] = ("achieved = algorithmic bytes (every gathered X row counted, SURVEY 8d) / kernel time; when X fits the 256 MiB "
                 "Infinity Cache most gathers never reach HBM and achieved may exceed the HBM peak -- frac is therefore taken "
                 "from the bytes that crossed the L2<->fabric boundary (PMC).  Since round 3 (XCD-affine column slices) the "
                 "headline moves a third fewer bytes across the fabric in 17 % less time: frac fell because the numerator did, "
                 "and the launch now sits against the L2 -> CU gather rate (block `l2`), not the fabric")
    if l2_request_bytes or all_hit_ms:
        # the other ceiling of a gather: with XCD-affine column slices most rows of the headline are served by L2, and the
        # launch is bound by the L2 -> CU path, not by what crosses the fabric
        r["l2"] = {"peak_gbs": L2_PEAK_GBS}
        if l2_request_bytes:
            r["l2"].update({"request_bytes": l2_request_bytes, "gbs": l2_request_bytes / t / 1e9,
                            "frac_of_peak": l2_request_bytes / t / 1e9 / L2_PEAK_GBS})
        if all_hit_ms:
            if cache_resident and all_hit_ms / case["kernel_ms"] >= 0.7:
                r["bound"] = "l2 gather rate (X resident in the Infinity Cache; the launch takes < 1.43x its all-hit time)"
            r["l2"].update({"all_hit_kernel_ms": all_hit_ms, "frac_of_all_hit_floor": all_hit_ms / case["kernel_ms"],
                            "all_hit_note": "the same rows and task schedule sizes with every column id folded into [0, 2048) (X = 256 KB per "
                                            "panel: every gather an L2 hit), timed in this run: what the launch would take if nothing missed L2"})
    if h.n_dense:
        flops = 2.0 * 16.0 * float(h.dense_k_sum) * D  # exactly what the MFMA chain of every dense window executes
        r["dense_path"] = {"flops_per_launch": flops, "tflops": flops / t / 1e12,
                           "frac_fp32_mfma_peak": flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                           "note": "lower bound on the dense-tile path's own rate: the whole launch time (sparse rows included) is charged to it"}
    return r


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child processes of this very command
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; nothing here has touched a GPU -- counting devices does not initialise
    one on this image), relay rank 0's JSON line, exit non-zero if any rank does."""
    import torch
    rehearsal = os.environ.get("HCSPMM_BENCH_REHEARSAL", "0") == "1"
    dry = os.environ.get("HCSPMM_BENCH_DRYRUN", "0") == "1"
    visible = torch.cuda.device_count()
    if visible < n and not (rehearsal or dry):
        print("bench.py: --gpus %d needs %d visible GPUs, found %d (HCSPMM_BENCH_REHEARSAL=1 runs every rank on GPU 0 over gloo "
              "to rehearse the plumbing)" % (n, n, visible), file=sys.stderr)
        return 3
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HCSPMM_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad or all(c is not None for c in codes):
                rc = bad[0] if bad else 0
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # a failed rank leaves the others waiting in a collective: stop exactly the processes started here
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    if rc != 0:
        print("bench.py: a rank exited with code %d" % rc, file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # SAG.profile's round count, GNN_model.py:251
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="reddit", choices=sorted(WORKLOADS) + sorted(STRONG_WORKLOADS))
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--frontend", default="extension", choices=["ctypes", "extension"],
                    help="Python front-end over the C ABI: the torch extension HCSPMM (the reference's boundary, hybrid_all.cpp:500-525; default) "
                         "or the ctypes glue; the line's `frontends` block times both")
    ap.add_argument("--no-loi", action="store_true", help="skip the `loi` block (community-structured graph: reorder x classifier)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="headline only (no dim 32 / 256, config 2 / 4 / 5 entries)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic then comes from profiles/ or is null)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"],
                    help="feature element type; f32 is the reference's (and BASELINE's) -- the 16-bit variants are the paper's Table VII extension")
    ap.add_argument("--rule", type=int, default=3, help="window classifier (hcspmm.h): 3 = MI355X refit, width-agnostic (the library's default), 4 = MI355X refit for "
                                                       "widths >= 64; the reference's: 0 intended, 1 guarded, 2 as shipped = all sparse")
    ap.add_argument("--no-plan", action="store_true", help="use the plan-free (reference-convention) kernel")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="one-GPU run of ONE rank's local product in a P-GPU job: the row block references columns of "
                         "all P blocks and X holds all P*n rows (already 'gathered'); no communication is timed (0 = the workload's own)")
    ap.add_argument("--strong-scale", type=int, default=1, help=argparse.SUPPRESS)  # tests: shrink a strong-scaling graph by this factor
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cases", default="", help=argparse.SUPPRESS)
    ap.add_argument("--graph-cache", default="", help="directory holding (or receiving) the generated graphs as <workload>_rp.npy / _col.npy")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launcher = os.environ.get("HCSPMM_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "none")
    if os.environ.get("HCSPMM_BENCH_DRYRUN", "0") == "1" and world > 1:
        # launcher self-test (tests/test_bench_cpu.py, no GPU): rendezvous over gloo, one collective, rank 0 prints a line
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"dryrun": True, "world_size": dist.get_world_size(), "sum_of_ranks_plus_one": t.item(),
                              "backend": dist.get_backend(), "launcher": launcher, "argv": sys.argv[1:]}))
        dist.destroy_process_group()
        return
    strong = args.workload in STRONG_WORKLOADS
    if strong:
        n_total, e_total, d_default, desc = STRONG_WORKLOADS[args.workload]
        vw_default = 1
    else:
        n_local, e_local, d_default, vw_default, desc = WORKLOADS[args.workload]
    args.dim = args.dim or d_default
    if args.virtual_world <= 0:
        args.virtual_world = vw_default
    D = args.dim
    vworld = max(1, args.virtual_world) if world == 1 else 1

    # ---- the graphs (host, numpy): made before anything touches the GPU so that the counter passes share them
    cache_dir = args.graph_cache
    own_cache = False
    do_sweep = world == 1 and not args.no_sweep and not strong and not args.pmc_child
    # (a strong-scaling graph on one GPU gets its counter passes too, up to 100 M entries: beyond that the passes' second copy
    # of the graph on the host and their run time are not worth a `traffic` figure)
    do_pmc = world == 1 and not args.pmc_child and not args.no_pmc and not (strong and STRONG_WORKLOADS[args.workload][1] // max(1, args.strong_scale) > 100_000_000)
    if do_pmc and not cache_dir:
        cache_dir = tempfile.mkdtemp(prefix="hcspmm_bench_")
        own_cache = True
    graphs_host = {}
    loi_info = {}
    do_loi = do_sweep and not args.no_loi and args.dtype == "f32" and not args.no_plan

    def graph_of(wl):
        """(rp, col) of workload `wl` as one GPU sees it: from the cache directory when it is there, else generated (and cached)."""
        if wl not in graphs_host:
            f_rp, f_col = (os.path.join(cache_dir, "%s_%s.npy" % (wl, n)) for n in ("rp", "col")) if cache_dir else (None, None)
            if cache_dir and os.path.exists(f_col):
                graphs_host[wl] = (np.load(f_rp), np.load(f_col))
            elif wl in STRONG_WORKLOADS:
                raise SystemExit("bench.py: strong-scaling graph %s is not in the graph cache" % wl)
            elif wl == "community_loi":
                graphs_host[wl] = reorder_for_loi_block(*graph_of("community"), loi_info)
                if cache_dir and world == 1:
                    np.save(f_rp, graphs_host[wl][0])
                    np.save(f_col, graphs_host[wl][1])
            else:
                nl, el, _, vw, _ = WORKLOADS[wl]
                w = world * vworld if wl == args.workload else vw
                graphs_host[wl] = make_local_block(wl, nl, el, w, rank)
                if cache_dir and world == 1:
                    os.makedirs(cache_dir, exist_ok=True)
                    np.save(f_rp, graphs_host[wl][0])
                    np.save(f_col, graphs_host[wl][1])
        return graphs_host[wl]

    if strong and args.pmc_child:
        rp, col = graph_of(args.workload)  # written by the parent
        n_local = n_total = len(rp) - 1
    elif strong:
        rp, col, n_local, n_total = make_strong_block(args.workload, world, rank, scale=max(1, args.strong_scale))
        if do_pmc:
            graphs_host[args.workload] = (rp, col)
            os.makedirs(cache_dir, exist_ok=True)
            np.save(os.path.join(cache_dir, "%s_rp.npy" % args.workload), rp)
            np.save(os.path.join(cache_dir, "%s_col.npy" % args.workload), col)
    else:
        rp, col = graph_of(args.workload)
    sweep_plan = [(w, d) for w, d in SWEEP_PLAN if not (w == args.workload and d == D)] if do_sweep else []
    cases = [(args.workload, D, args.rule)] + [(w, d, DEFAULT_RULE) for w, d in sweep_plan] + (loi_plan() if do_loi else [])
    if args.pmc_child and args.cases:
        cases = [(c.split(":")[0], int(c.split(":")[1]), int((c.split(":") + ["0"])[2])) for c in args.cases.split(",")]
    pmc = None
    import torch  # (importing does not initialise the GPU; doing it first pages the library in for the child processes too)
    if do_loi:
        graph_of("community_loi")  # generated and reordered on the host before anything touches the GPU (the reorder is timed here)
    if do_pmc:
        for c in cases:
            graph_of(c[0])
        passes = [p for p in os.environ.get("HCSPMM_BENCH_PMC", "fetch,write,l2,mfma").split(",") if p in PMC_PASSES]
        pmc = live_pmc(args, cache_dir, cases, passes)

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # HCSPMM_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo collectives -- lets the N > 1 plumbing run
    # on a one-GPU box (numbers from it are meaningless and are labelled as such)
    rehearsal = os.environ.get("HCSPMM_BENCH_REHEARSAL", "0") == "1"
    visible = torch.cuda.device_count()
    if world > 1 and not rehearsal and visible < world:
        print("bench.py: %d ranks but only %d visible GPUs" % (world, visible), file=sys.stderr)
        sys.exit(3)
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

    fe = _Frontend(args.frontend)
    n_gather_panels = int(os.environ.get("HCSPMM_GATHER_PANELS", "0"))
    if args.pmc_child:
        # one process, every case in turn, ONE preprocess each (its edge_to_row_kernel launch marks the case in the counter output)
        for i, (w, d, r) in enumerate(cases):
            rp_c, col_c = graph_of(w)
            nl, vw = (len(rp_c) - 1, 1) if w in STRONG_WORKLOADS else (WORKLOADS[w][0], WORKLOADS[w][3])
            head = i == 0 and w == args.workload
            run_case(fe, dev, w, d, rp_c, col_c, nl, 1, 0, vworld if head else vw, args.steps, args.warmup,
                     args.dtype if head else "f32", args.rule if head else r, args.no_plan if head else False, 0, None, prep_runs=1)
            torch.cuda.empty_cache()
        return
    case = run_case(fe, dev, args.workload, D, rp, col, n_local, world, rank, vworld, args.steps, args.warmup, args.dtype,
                    args.rule, args.no_plan, n_gather_panels, dist)
    elapsed, E, header = case["elapsed"], case["E"], case["header"]

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
        key = case_key(args.workload, D, args.dtype)
        traffic, source, extra = None, None, {}
        if pmc and key in pmc and "traffic_bytes" in pmc[key]:
            traffic = pmc[key]["traffic_bytes"]
            source = ("this run: rocprofv3 --pmc child passes of bench.py on the same graph before the timed region "
                      "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB); seconds per pass (all cases) %s" % json.dumps(pmc["_passes"]))
            extra = {k: pmc[key].get(k) for k in ("fetch_bytes", "write_bytes", "l2_hit_rate", "mfma_util_percent")}
        elif world == 1 and args.rule == DEFAULT_RULE and not args.no_plan:
            rec = recorded_profile(key)
            if rec and not rec.get("stale"):
                traffic = rec.get("traffic_bytes")
                source = "recorded: %s (builder rocprofv3 run, kernel_src_sha %s)" % (rec.get("source"), rec.get("kernel_src_sha"))
                extra = {"l2_hit_rate": rec.get("l2_hit_rate")}
        all_hit_ms = None
        if world == 1 and not strong and not args.no_sweep and not args.no_plan and len(col) <= 64_000_000:
            try:  # the L2 gather floor of this degree sequence (a few seconds: one sort on the host, 20 steps)
                floor = run_case(fe, dev, args.workload, D, rp, fold_columns(rp, col), n_local, 1, 0, vworld, 20, 5, args.dtype, 2,
                                 False, 0, None, prep_runs=1)
                all_hit_ms = floor["kernel_ms"]
                del floor
            except Exception:  # reported as absent, never fatal
                all_hit_ms = None
        roof = roofline_of(case, traffic, source, (pmc or {}).get(key, {}).get("l2_request_bytes") if pmc else None, all_hit_ms)
        roof.update({k: v for k, v in extra.items() if v is not None})
        if pmc and "error" in pmc:
            roof["pmc_error"] = pmc["error"]
        if strong:
            workload_txt = ("%s: %s; %d nodes / %d stored entries in all, %d / %d on rank 0, dim %d"
                            % (args.workload, desc, n_total, int(total_edges), n_local, E, D))
        else:
            workload_txt = ("%s: %s; %d nodes / %d stored entries per GPU, dim %d%s"
                            % (args.workload, desc, n_local, E, D,
                               "" if vworld == 1 else "; ONE rank of a virtual %d-GPU job (X: %d rows resident)" % (vworld, n_local * vworld)))
        out = {
            "metric": "GNN-aggregation SpMM edges*dim/s (A*X, %s)" % ("fp32" if args.dtype == "f32" else args.dtype + " features, fp32 accumulation"),
            "value": total_edges * D / (elapsed / args.steps),
            "unit": "edge*dim/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)" if rehearsal else ""),
            "config": {"workload": workload_txt,
                       "nodes_per_gpu": n_local, "entries_per_gpu": E, "dim": D,
                       "parallelism": "row-block shard x%d + all-gather(X) in %d column panel(s), panel-major persistent buffers"
                                      % (world, case["n_gather_panels"] if world > 1 else 1),
                       "frontend": args.frontend, "plan": (not args.no_plan), "rule": args.rule, "sparse_tasks": header.n_tasks,
                       "dense_windows": header.n_dense, "split_rows": header.n_split_rows,
                       "column_slices": getattr(header, "n_slices", 0), "slice_threshold": getattr(header, "slice_threshold", 0),
                       "preprocess_ms": case["prep_warm_ms"], "preprocess_ms_cold": case["prep_cold_ms"],
                       "kernel_src_sha": kernel_src_sha()},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1,
                            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else None,
                            "visible_gpus": visible, "launcher": launcher},
            "roofline": roof,
        }
        if world > 1:
            # the one exchange of the path: every rank receives the other ranks' X row blocks each step.  On the 8-GPU full
            # mesh (7 xGMI links x 153.6 GB/s bidirectional per GPU) an all-gather is bound by ONE link per peer.
            recv = float(world - 1) * n_local * D * case["elem"]
            out["communication"] = {"collective": "all_gather_into_tensor (RCCL) of X, %d column panel(s), panel-major buffers, no copies"
                                                  % case["n_gather_panels"],
                                    "bytes_received_per_rank_per_step": recv,
                                    "one_link_per_peer_bound_ms": n_local * D * case["elem"] / 76.8e9 * 1e3,
                                    "local_product_ms": case["kernel_ms"],
                                    "exposed_ms": ms_per_step - case["kernel_ms"],
                                    "note": "communication-bound by construction (DESIGN.md section 6): the gathered bytes grow with the "
                                            "world size while the local product does not; only the last panel's product is exposed"}
        torch.cuda.empty_cache()
        if sweep_plan:
            out["sweep"] = sweep(fe, dev, args, sweep_plan, graph_of, pmc)
        if sweep_plan and any(w == "rd_like" for w, _ in sweep_plan):
            out["fused"] = fused_block(dev, graph_of)
        if world == 1 and not strong and not args.no_sweep:
            out["frontends"] = frontends_block(fe, dev, args, rp, col, n_local, vworld, D, ms_per_step, case["kernel_ms"])
        if do_loi:
            out["loi"] = loi_block(fe, dev, args, graph_of, pmc, loi_info)
        if world == 1 and not args.no_cpu_baseline and not strong:
            Xh = torch.randn(case["x_rows"], D, generator=torch.Generator().manual_seed(1234)).numpy()
            out["cpu_baseline"] = cpu_baseline(rp, col, Xh, D, case["x_rows"])
        print(json.dumps(out))
    if own_cache:
        shutil.rmtree(cache_dir, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def reorder_for_loi_block(rp, col, info):
    """Host side of the `loi` block, before the GPU is touched: the relaxed parallel LOI reorder of the (shuffled) community graph
    (timed: cold = first call of the process, then the best of two more), the permutation applied, and -- for scale -- the exact
    reorder_plus_new_direct restatement on one core.  -> (row_pointers, column_index) of the reordered graph; timings into `info`."""
    import torch
    import hcspmm
    rpt, colt = torch.from_numpy(rp), torch.from_numpy(col)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        perm, sizes = hcspmm.loi_reorder(rpt, colt, variant="fast")
        times.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    rpr, colr = hcspmm.apply_permutation(rpt, colt, perm)
    t_apply = time.perf_counter() - t0
    info.update({"variant": "fast (hcspmm_loi_reorder_fast: capped list walks, deterministic reservation rounds; NOT the reference's permutation)",
                 "reorder_s": min(times[1:]), "reorder_cold_s": times[0], "apply_permutation_s": t_apply,
                 "groups": int(sizes.numel()), "full_groups": int((sizes == 16).sum()),
                 "host_threads": "min(16, HCSPMM_THREADS or hardware) on the caller's L3 domain (HCSPMM_LOI_PIN)", "host_cores": os.cpu_count()})
    if os.environ.get("HCSPMM_BENCH_LOI_EXACT", "1") == "1":
        t0 = time.perf_counter()
        _, sizes_x = hcspmm.loi_reorder(rpt, colt)
        info["exact"] = {"variant": "reorder_plus_new_direct (LOI.cpp:660-805), bit-exact restatement, one core",
                         "reorder_s": time.perf_counter() - t0, "groups": int(sizes_x.numel()), "full_groups": int((sizes_x == 16).sum())}
    return rpr.numpy(), colr.numpy()


def oracle_sample_check(rp, col, X, Z, n_rows=4096, seed=0):
    """A sample of rows (random ones, the first 256 -- whole windows -- and the 64 longest) of Z = A*X against the CPU oracle's
    fp32 CSR-order product; only the X rows those reference leave the GPU.  -> {"rows", "ok", "worst_over_1e-5_bar"}."""
    import torch
    import oracle
    N = len(rp) - 1
    deg_all = np.diff(rp)
    rng = np.random.default_rng(seed)
    idx = np.unique(np.concatenate([rng.integers(0, N, n_rows), np.arange(min(N, 256)), np.argsort(deg_all)[-64:]])).astype(np.int64)
    deg = deg_all[idx].astype(np.int64)
    rp_s = np.concatenate([[0], np.cumsum(deg)])
    pos = np.repeat(rp[idx].astype(np.int64) - rp_s[:-1], deg) + np.arange(rp_s[-1])
    col_s = col[pos].astype(np.int64)
    used, inv = np.unique(col_s, return_inverse=True)
    Xs = X[torch.from_numpy(used).to(X.device)].float().cpu().numpy()
    Zs = Z[torch.from_numpy(idx).to(Z.device)].float().cpu().numpy()
    ok, worst = oracle.check_spmm(Zs, rp_s.astype(np.int32), inv.astype(np.int32), Xs)
    return {"rows": int(idx.shape[0]), "ok": bool(ok), "worst_over_1e-5_bar": float(worst)}


def gcn_epoch_ms(dev, rp, col, epochs=8):
    """One training epoch (forward + backward + Adam) of the reference's default GCN (HC-SpMM_main.py:19-25: 6 layers, dim 96, hidden 32,
    22 classes) through the reference's own boundary (the HCSPMM extension + GNN_model.py), milliseconds."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    pkg = os.path.join(ROOT, "hc-spmm_amd")
    for q in (pkg, os.path.join(pkg, "hybrid_kernel")):
        if q not in sys.path:
            sys.path.insert(0, q)
    import HCSPMM
    from GNN_model import GCNConv
    n = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    graph = (rp_d, col_d, *HCSPMM.preprocess(col_d, rp_d, n, len(col), (n + 15) // 16))
    x = torch.randn(n, 96, device=dev)
    y = torch.ones(n, dtype=torch.long, device=dev)
    output = torch.zeros(n, 32, device=dev)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = GCNConv(96, 32, 1)
            self.hidden_layers = nn.ModuleList(GCNConv(32, 32, 0) for _ in range(4))
            self.conv2 = GCNConv(32, 22, 2)

        def forward(self):
            h = F.relu(self.conv1(x, *graph, output))
            h = F.dropout(h, training=self.training)
            for c in self.hidden_layers:
                h = F.relu(c(h, *graph, output))
            return F.log_softmax(self.conv2(h, *graph, output), dim=1)
    net = Net().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=0.01)

    def train():
        net.train()
        opt.zero_grad()
        loss = -net().gather(1, y.unsqueeze(1)).mean()
        loss.backward()
        opt.step()
    for _ in range(4):
        train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        train()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / epochs * 1e3


def loi_block(fe, dev, args, graph_of, pmc, info):
    """The reference's second contribution on the scoreboard: the LOI layout reorder feeding the window classifier feeding the two
    sub-paths (LOI.cpp:660-805 -> hybrid_all_kernel.cu:261-262 -> :960-1121; paper p.7, p.13-14).  Workload: a community-structured
    graph of the paper's RD size whose structure is hidden by a random vertex numbering.  For {as shuffled, after the reorder} x
    {rule 0 = the reference's classifier (fitted on an RTX 3090), the MI355X refit for the width}: time per SpMM, windows on the
    dense-tile path, bytes across the fabric, L2 hit rate and MFMA-busy share from this run's own counter passes, and a sampled-row
    check against the CPU oracle; plus what the reorder costs against 200 training epochs.  Never part of `value`."""
    import torch
    out = {"workload": WORKLOADS["community"][4], "reorder": dict(info), "cases": []}
    steps, warmup = max(10, min(args.steps, 30)), 5
    by_key = {}
    try:
        for wl, D, rule in loi_plan():
            rp, col = graph_of(wl)
            n = len(rp) - 1
            case = run_case(fe, dev, wl, D, rp, col, n, 1, 0, 1, steps, warmup, "f32", rule, keep_op=True)
            op = case.pop("op")
            h = case["header"]
            check = oracle_sample_check(rp, col, op.X_pm[0], op.Z_pm[0])
            live = (pmc or {}).get(case_key(wl, D, "f32", rule)) if pmc and "error" not in pmc else None
            e = {"graph": "as shuffled" if wl == "community" else "after LOI reorder", "dim": D, "rule": rule,
                 "classifier": {0: "reference (RTX 3090 fit, hybrid_all_kernel.cu:261)", 2: "none: every window on the sparse-row path (hybrid_all_kernel.cu:262 as shipped)",
                                3: "MI355X refit (narrow)", 4: "MI355X refit (wide)"}[rule],
                 "nodes": n, "entries": case["E"], "kernel_ms": case["kernel_ms"], "ms_per_step": case["elapsed"] / steps * 1e3,
                 "value": case["E"] * D / (case["elapsed"] / steps), "unit": "edge*dim/s",
                 "dense_windows": h.n_dense, "windows": (n + 15) // 16, "dense_window_share": h.n_dense / max((n + 15) // 16, 1),
                 "entries_on_dense_path_share": h.nnz_dense / max(case["E"], 1), "unique_columns_gathered_by_dense_windows": h.uniq_dense,
                 "preprocess_ms": case["prep_warm_ms"], "oracle_check": check}
            for k_src, k_dst in (("traffic_bytes", "fabric_bytes"), ("l2_hit_rate", "l2_hit_rate"), ("mfma_util_percent", "mfma_busy_percent"),
                                 ("fetch_bytes", "fetch_bytes"), ("write_bytes", "write_bytes")):
                if live and live.get(k_src) is not None:
                    e[k_dst] = live[k_src]
            if live and live.get("traffic_bytes"):
                e["fabric_gbs"] = live["traffic_bytes"] / (case["kernel_ms"] * 1e-3) / 1e9
                e["frac_hbm_peak"] = e["fabric_gbs"] / HBM_PEAK_GBS
                e["compulsory_bytes"] = compulsory_bytes(n, case["E"], D, case["cols_referenced"])
            out["cases"].append(e)
            by_key[(wl, D, rule)] = e
            del case, op
            torch.cuda.empty_cache()
        summary = {}
        for D in (128, 32):
            r_mi = mi355x_rule(D)
            a0, a1 = by_key[("community", D, 0)], by_key[("community", D, r_mi)]
            b0, b1 = by_key[("community_loi", D, 0)], by_key[("community_loi", D, r_mi)]
            summary["dim%d" % D] = {
                "shuffled_ms": {"rule0": a0["kernel_ms"], "mi355x": a1["kernel_ms"]},
                "reordered_ms": {"rule0": b0["kernel_ms"], "mi355x": b1["kernel_ms"]},
                "gain_from_reorder_percent": {"rule0": 100.0 * (a0["kernel_ms"] - b0["kernel_ms"]) / a0["kernel_ms"],
                                              "mi355x": 100.0 * (a1["kernel_ms"] - b1["kernel_ms"]) / a1["kernel_ms"]},
                "mi355x_rule_vs_rule0_percent": {"shuffled": 100.0 * (a0["kernel_ms"] - a1["kernel_ms"]) / a0["kernel_ms"],
                                                 "reordered": 100.0 * (b0["kernel_ms"] - b1["kernel_ms"]) / b0["kernel_ms"]}}
            if "fabric_bytes" in a0 and "fabric_bytes" in b0:
                summary["dim%d" % D]["fabric_bytes"] = {"shuffled_rule0": a0["fabric_bytes"], "reordered_rule0": b0["fabric_bytes"]}
            s2 = by_key.get(("community_loi", D, 2))
            if s2:
                summary["dim%d" % D]["reordered_all_sparse_ms"] = s2["kernel_ms"]
                summary["dim%d" % D]["dense_tile_path_gain_percent"] = 100.0 * (s2["kernel_ms"] - b1["kernel_ms"]) / s2["kernel_ms"]
        out["summary"] = summary
        if os.environ.get("HCSPMM_BENCH_LOI_EPOCHS", "1") == "1":
            ep = {}
            for wl in ("community", "community_loi"):
                ep["as shuffled" if wl == "community" else "after LOI reorder"] = gcn_epoch_ms(dev, *graph_of(wl))
                torch.cuda.empty_cache()
            cost = info.get("reorder_s", float("nan")) + info.get("apply_permutation_s", 0.0)
            run200 = 200 * ep["as shuffled"] * 1e-3
            out["training"] = {"model": "GCN, 6 layers, dim 96, hidden 32, 22 classes (HC-SpMM_main.py:19-25), forward + backward + Adam through the HCSPMM extension",
                               "epoch_ms": ep, "epochs_200_s_as_shuffled": run200,
                               "reorder_plus_apply_s": cost, "reorder_cost_percent_of_200_epochs": 100.0 * cost / run200,
                               "reorder_alone_percent_of_200_epochs": 100.0 * info.get("reorder_s", float("nan")) / run200,
                               "epochs_to_amortise": cost / max((ep["as shuffled"] - ep["after LOI reorder"]) * 1e-3, 1e-12),
                               "paper": "LOA cost 6.58 % of a 200-epoch run, +8.40 % average (p.13-14, RTX 3090)"}
        out["default_rule"] = ("preprocess defaults to rule 3, the MI355X refit that holds at every embedding width (never slower than the reference's "
                               "RTX-3090 coefficients on any workload measured, up to 25 % faster: profiles/r04/ab_classifier_rules.log); "
                               "hcspmm.preprocess(rule='mi355x', dim=D) / HCSPMM.set_rule(4) pick the wide set for D >= 64; HCSPMM.set_rule(0) / "
                               "HCSPMM_RULE=0 select the reference's coefficients (hybrid_type then equals the reference's bit for bit) -- both are timed above")
    except Exception as ex:  # never takes the headline down
        out["error"] = str(ex)[:400]
    return out


def frontends_block(fe, dev, args, rp, col, n_local, vworld, D, ms_headline, kernel_ms_headline, steps=50):
    """The headline workload through the OTHER Python front-end over the same C ABI: the reference's boundary is the HCSPMM torch
    extension (hybrid_all.cpp:500-525), the ctypes glue needs no compiler; both are parity-tested, the headline runs on --frontend."""
    other = "ctypes" if fe.name == "extension" else "extension"
    try:
        fo = _Frontend(other)
        c = run_case(fo, dev, args.workload, D, rp, col, n_local, 1, 0, vworld, steps, 10, args.dtype, args.rule, args.no_plan, prep_runs=1)
        ms_other = c["elapsed"] / steps * 1e3
        # ... and the headline's front-end once more, same step count: how much two runs of ONE front-end differ (a new plan
        # and new buffers land at other addresses) stands beside how much the two front-ends do
        a = run_case(fe, dev, args.workload, D, rp, col, n_local, 1, 0, vworld, steps, 10, args.dtype, args.rule, args.no_plan, prep_runs=1)
        ms_again = a["elapsed"] / steps * 1e3
        return {fe.name: {"ms_per_step": ms_headline, "kernel_ms": kernel_ms_headline, "steps": args.steps, "headline": True},
                other: {"ms_per_step": ms_other, "kernel_ms": c["kernel_ms"], "steps": steps, "headline": False},
                fe.name + "_again": {"ms_per_step": ms_again, "kernel_ms": a["kernel_ms"], "steps": steps, "headline": False},
                "difference_percent": 100.0 * (ms_other - ms_again) / ms_again,
                "difference_basis": "%s against %s_again (same step count, back to back)" % (other, fe.name)}
    except Exception as ex:
        return {"error": str(ex)[:300]}


def fused_block(dev, graph_of, wl="rd_like", D=32, H=32, steps=30):
    """The fused aggregate+update operator (the GCN-backward shape) on the RD-sized low-degree graph of the sweep:
    two launches (a plan built with fuse_in_launch = -1) against the row-tile form the operator picks by itself when the
    aggregate is 80 MB or more -- HIP events around `steps` calls each, same graph, same tensors, out compared bit for bit.
    Not part of `value`; never takes the headline down."""
    import torch
    import hcspmm
    try:
        rp, col = graph_of(wl)
        N, E = len(rp) - 1, len(col)
        rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
        outs = hcspmm.preprocess(col_d, rp_d, N, E, (N + 15) // 16)
        never = hcspmm.build_plan(rp_d, col_d, outs[0], outs[1], outs[3], fuse_in_launch=-1)
        X, W = torch.randn(N, D, device=dev), torch.randn(D, H, device=dev)

        def timed(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(steps):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / steps
        a_auto = (rp_d, col_d, *outs)
        a_two = (rp_d, col_d, outs[0], outs[1], outs[2], outs[3], never, outs[5])
        form = int(hcspmm.fused_in_launch(outs[4], D, H))
        t_two = timed(lambda: hcspmm.forward_fixed32_fused(X, *a_two, W))
        t_auto = timed(lambda: hcspmm.forward_fixed32_fused(X, *a_auto, W))
        t_spmm = timed(lambda: hcspmm.forward(X, *a_auto))
        o_two, o_auto = hcspmm.forward_fixed32_fused(X, *a_two, W), hcspmm.forward_fixed32_fused(X, *a_auto, W)
        same = bool(torch.equal(o_two[0], o_auto[0]) and torch.equal(o_two[1], o_auto[1]))
        h = hcspmm.plan_header(outs[4])
        res = {"operator": "forward_fixed32_fused: out2 = A*X, out = out2*W", "workload": wl, "nodes": N, "entries": E, "dim": D, "hidden": H,
               "dense_windows": h.n_dense, "sparse_tasks": h.n_tasks, "two_launches_ms": t_two, "chosen_form": form,
               "chosen_form_ms": t_auto, "gain_percent": 100.0 * (t_two - t_auto) / t_two, "spmm_alone_ms": t_spmm,
               "out_and_out2_bit_identical_between_forms": same, "steps": steps,
               "note": "form 2 = 16-row tiles of both sub-paths summed, parked in LDS and multiplied by W before they leave the CU "
                       "(csrc/fused_rows.hip); chosen automatically when out2 is 80 MB or more at dim <= 64"}
        del outs, never, X, W, o_two, o_auto
        torch.cuda.empty_cache()
        return res
    except Exception as ex:
        return {"error": str(ex)[:300]}


def sweep(fe, dev, args, plan, graph_of, pmc):
    """The other BASELINE points, timed in this process after the headline (fewer steps each); traffic / L2 hit rate /
    MFMA utilisation of each from this run's own counter passes (`pmc`), the recorded table only when those failed."""
    import torch
    entries = []
    steps, warmup = max(10, min(args.steps, 50)), 5
    for wl, D in plan:
        n_local, e_local, _, vw, desc = WORKLOADS[wl]
        t0 = time.perf_counter()
        try:
            rp, col = graph_of(wl)
            gen_s = time.perf_counter() - t0
            case = run_case(fe, dev, wl, D, rp, col, n_local, 1, 0, vw, steps, warmup, "f32", DEFAULT_RULE)
            key = case_key(wl, D, "f32", DEFAULT_RULE)
            live = pmc.get(key) if pmc and "error" not in pmc else None
            traffic, src, rec = None, None, None
            if live and live.get("traffic_bytes"):
                traffic = live["traffic_bytes"]
                src = "this run: rocprofv3 --pmc child passes of bench.py on the same graph before the timed region"
            else:
                rec = recorded_profile(key)
                traffic = rec.get("traffic_bytes") if rec else None
                if traffic:
                    src = "recorded: %s%s%s" % (rec.get("source"), " (STALE: kernel sources changed since)" if rec.get("stale") else "",
                                                 "; live counter passes failed: " + pmc["error"] if pmc and "error" in pmc else "")
            roof = roofline_of(case, traffic, src)
            h = case["header"]
            e = {"workload": wl, "dim": D, "nodes": n_local, "entries": case["E"], "x_rows_resident": case["x_rows"],
                 "steps": steps, "ms_per_step": case["elapsed"] / steps * 1e3, "kernel_ms": case["kernel_ms"],
                 "value": case["E"] * D / (case["elapsed"] / steps), "unit": "edge*dim/s",
                 "sparse_tasks": h.n_tasks, "dense_windows": h.n_dense, "nnz_dense": h.nnz_dense, "split_rows": h.n_split_rows,
                 "column_slices": getattr(h, "n_slices", 0),
                 "preprocess_ms": case["prep_warm_ms"], "graph_gen_s": round(gen_s, 1), "roofline": roof, "desc": desc}
            if case.get("hip_graph_ms_per_step") is not None:
                e["hip_graph_ms_per_step"] = case["hip_graph_ms_per_step"]
                e["hip_graph_note"] = "the same step captured in a HIP graph and replayed: launch-bound workload, the Python call costs more than the kernel"
            for k in ("l2_hit_rate", "mfma_util_percent", "fetch_bytes", "write_bytes"):
                v = (live or {}).get(k) if live else (rec or {}).get(k)
                if v is not None:
                    e[k] = v
            if rec:
                e["profile_source"] = rec.get("source")
                e["profile_stale"] = rec.get("stale")
            entries.append(e)
            del case
        except Exception as ex:  # a sweep entry never takes the headline down
            entries.append({"workload": wl, "dim": D, "error": str(ex)[:300]})
        torch.cuda.empty_cache()
    return entries


if __name__ == "__main__":
    main()
