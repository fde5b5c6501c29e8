#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hybrid SpMM hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--dim D] [--frontend ctypes|extension]

A "step" is one pass of the hot path, Z = A*X, over one synthetic batch (graph + embedding matrix)
already resident in HBM.  N = 1 runs BASELINE.json config 3, "Reddit-scale": 233 000 nodes /
11.6 M stored entries, power-law, dim 128 (the configuration the metric "dim=128" is quoted on that
fits one GPU).  N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL) is weak scaling:
every rank owns one such row block (columns span all N*233 000 vertices) and each step all-gathers
the embedding row blocks over xGMI before its local product -- the one exchange step the path has.

Prints ONE JSON line on rank 0 (contract in the task description).  At N = 1 the same line carries
  * `roofline`: live HIP-event kernel time, and -- measured by THIS run -- the bytes that crossed the fabric
    per launch: before the process touches the GPU it runs itself under `rocprofv3 --pmc` (FETCH_SIZE,
    WRITE_SIZE, TCC hit/miss: separate passes, as MI355X_MICROARCH.md prescribes) on the same graph and reads
    the counters back.  `frac` is that traffic / time / 8 TB/s (<= 1 by construction); the algorithmic and
    compulsory figures stand beside it.  If the profiler is unavailable the recorded figure under profiles/
    is used and labelled with its source; failing that, the compulsory-bytes fraction.
  * `sweep`: the other BASELINE points timed in the same process -- Reddit-scale at dim 32 and 256, the
    Cora-scale config 2, one GPU's share of config 4 (dim 256) and one GPU's share of config 5 as SURVEY.md 8(d)
    defines it (planted <= 24-column groups over 16 M columns: most windows on the dense-tile / MFMA path);
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
    "dense": (2000000, 0, 128, 1, "2 M-node square graph, 70 % planted windows of 20 columns + 16 random entries per other row (round-1 dense-heavy proxy)"),
}

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
        return graphs.powerlaw_graph(n_local, e_local, seed=seed)
    return graphs.powerlaw_block(n_local, n_local * world, e_local, seed=seed, rank=rank)


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
            self.m.set_rule(int(rule))
            try:
                return self.m.preprocess(col_d, rp_d, N, E, W, int(num_columns))
            finally:
                self.m.set_rule(0)
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
PMC_PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"], "l2": ["TCC_HIT_sum", "TCC_MISS_sum"]}


def _read_counters(d):
    import collections
    import csv
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "hcspmm::" in r.get("Kernel_Name", ""):
                    acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def live_pmc(args, cache_dir, passes):
    """-> dict with per-step counters of the headline launch, or {"error": ...}.  Each pass is its own run of
    `rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --pmc-child ...` (the program directly after `--`)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    # already inside a profiler (this very command run under rocprofv3, e.g. by tools/profile_round.sh or a harness)?  Then no
    # nested profiler runs: the recorded figures are used instead
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "already running under a profiler: live counter passes skipped"}
    out = {"passes": {}, "counters": {}}
    child = ["python3", os.path.join(ROOT, "bench.py"), "--pmc-child", "--graph-cache", cache_dir, "--workload", args.workload,
             "--dim", str(args.dim), "--steps", "4", "--warmup", "2", "--rule", str(args.rule), "--dtype", args.dtype,
             "--frontend", args.frontend, "--virtual-world", str(args.virtual_world)] + (["--no-plan"] if args.no_plan else [])
    env = dict(os.environ, TMPDIR="/tmp")
    for name in passes:
        d = os.path.join(cache_dir, "pmc_" + name)
        cmd = [rocprof, "--pmc"] + PMC_PASSES[name] + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
        t0 = time.perf_counter()
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.wait()
                return {"error": "rocprofv3 pass '%s' timed out" % name}
            if p.returncode != 0:
                return {"error": "rocprofv3 pass '%s' exited %d: %s" % (name, p.returncode, log.decode(errors="replace")[-300:])}
        except Exception as e:  # profiler trouble is reported, never fatal
            return {"error": "rocprofv3 pass '%s': %s" % (name, str(e)[:200])}
        got = _read_counters(d)
        if not got:
            return {"error": "rocprofv3 pass '%s' produced no counters for hcspmm kernels" % name}
        out["passes"][name] = round(time.perf_counter() - t0, 1)
        for (k, c), v in got.items():
            out["counters"].setdefault(c, {})[k] = v
    c = out["counters"]
    # MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
    # of wide (16 B / lane) reads -> doubled; WRITE_SIZE is exact for 16-byte stores.  Summed over the launches of a
    # step (hybrid kernel + fix-up); tools/calibrate_fetch_size.py checks the factor on a copy of known size.
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["fetch_bytes"] = 2.0 * 1024.0 * sum(c["FETCH_SIZE"].values())
        out["write_bytes"] = 1024.0 * sum(c["WRITE_SIZE"].values())
        out["traffic_bytes"] = out["fetch_bytes"] + out["write_bytes"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        hit, miss = sum(c["TCC_HIT_sum"].values()), sum(c["TCC_MISS_sum"].values())
        out["l2_hit_rate"] = hit / max(hit + miss, 1.0)
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
             no_plan=False, n_gather_panels=0, dist=None):
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
    for _ in range(2):  # cold (first touch: pinned staging buffers, code objects, allocator growth), then warm
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
    return {"workload": workload, "D": D, "N": n_local, "E": E, "n_cols": n_cols, "elem": elem, "header": header,
            "elapsed": elapsed, "steps": steps, "kernel_ms": kern_ms, "prep_cold_ms": prep[0], "prep_warm_ms": prep[1],
            "n_gather_panels": n_gather_panels, "cols_referenced": int(flags.sum()), "x_rows": n_cols,
            "hip_graph_ms_per_step": graph_ms}


def roofline_of(case, traffic=None, traffic_source=None):
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
                 "from the bytes that crossed the L2<->fabric boundary (PMC), which is what bounds the launch")
    if h.n_dense:
        flops = 2.0 * 16.0 * float(h.dense_k_sum) * D  # exactly what the MFMA chain of every dense window executes
        r["dense_path"] = {"flops_per_launch": flops, "tflops": flops / t / 1e12,
                           "frac_fp32_mfma_peak": flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                           "note": "lower bound on the dense-tile path's own rate: the whole launch time (sparse rows included) is charged to it"}
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # SAG.profile's round count, GNN_model.py:251
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="reddit", choices=sorted(WORKLOADS))
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--frontend", default="ctypes", choices=["ctypes", "extension"],
                    help="Python front-end over the C ABI: the ctypes glue, or the torch extension HCSPMM (the reference's boundary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="headline only (no dim 32 / 256, config 2 / 4 / 5 entries)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (traffic then comes from profiles/ or is null)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"],
                    help="feature element type; f32 is the reference's (and BASELINE's) -- the 16-bit variants are the paper's Table VII extension")
    ap.add_argument("--rule", type=int, default=0, help="window classifier (hcspmm.h: 0 intended, 2 as shipped = all sparse, 3 / 4 MI355X refits)")
    ap.add_argument("--no-plan", action="store_true", help="use the plan-free (reference-convention) kernel")
    ap.add_argument("--virtual-world", type=int, default=0,
                    help="one-GPU run of ONE rank's local product in a P-GPU job: the row block references columns of "
                         "all P blocks and X holds all P*n rows (already 'gathered'); no communication is timed (0 = the workload's own)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-cache", default="", help="directory holding (or receiving) the generated headline graph as rp.npy / col.npy")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
              % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    n_local, e_local, d_default, vw_default, desc = WORKLOADS[args.workload]
    args.dim = args.dim or d_default
    if args.virtual_world <= 0:
        args.virtual_world = vw_default
    D = args.dim
    vworld = max(1, args.virtual_world) if world == 1 else 1

    # ---- the headline graph (host, numpy): made before anything touches the GPU so that the counter passes share it
    cache_dir = args.graph_cache
    own_cache = False
    if cache_dir and os.path.exists(os.path.join(cache_dir, "col.npy")):
        rp, col = np.load(os.path.join(cache_dir, "rp.npy")), np.load(os.path.join(cache_dir, "col.npy"))
    else:
        rp, col = make_local_block(args.workload, n_local, e_local, world * vworld, rank)
        if cache_dir and world == 1:  # tools/profile_round.sh: the five passes of one workload share one generated graph
            os.makedirs(cache_dir, exist_ok=True)
            np.save(os.path.join(cache_dir, "rp.npy"), rp)
            np.save(os.path.join(cache_dir, "col.npy"), col)
    pmc = None
    import torch  # (importing does not initialise the GPU; doing it first pages the library in for the child processes too)
    if world == 1 and not args.pmc_child and not args.no_pmc:
        cache_dir = tempfile.mkdtemp(prefix="hcspmm_bench_")
        own_cache = True
        np.save(os.path.join(cache_dir, "rp.npy"), rp)
        np.save(os.path.join(cache_dir, "col.npy"), col)
        passes = [p for p in os.environ.get("HCSPMM_BENCH_PMC", "fetch,write,l2").split(",") if p in PMC_PASSES]
        pmc = live_pmc(args, cache_dir, passes)

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

    fe = _Frontend(args.frontend)
    n_gather_panels = int(os.environ.get("HCSPMM_GATHER_PANELS", "0"))
    case = run_case(fe, dev, args.workload, D, rp, col, n_local, world, rank, vworld, args.steps, args.warmup, args.dtype,
                    args.rule, args.no_plan, n_gather_panels, dist)
    if args.pmc_child:
        return
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
        key = "%s_d%d%s" % (args.workload, D, "" if args.dtype == "f32" else "_" + args.dtype)
        traffic, source, extra = None, None, {}
        if pmc and "traffic_bytes" in pmc:
            traffic = pmc["traffic_bytes"]
            source = ("this run: rocprofv3 --pmc child passes of bench.py on the same graph before the timed region "
                      "(FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KiB); seconds per pass %s" % json.dumps(pmc["passes"]))
            extra = {"fetch_bytes": pmc.get("fetch_bytes"), "write_bytes": pmc.get("write_bytes"), "l2_hit_rate": pmc.get("l2_hit_rate")}
        elif world == 1 and args.rule == 0 and not args.no_plan:
            rec = recorded_profile(key)
            if rec and not rec.get("stale"):
                traffic = rec.get("traffic_bytes")
                source = "recorded: %s (builder rocprofv3 run, kernel_src_sha %s)" % (rec.get("source"), rec.get("kernel_src_sha"))
                extra = {"l2_hit_rate": rec.get("l2_hit_rate")}
        roof = roofline_of(case, traffic, source)
        roof.update({k: v for k, v in extra.items() if v is not None})
        if pmc and "error" in pmc:
            roof["pmc_error"] = pmc["error"]
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
                       "nodes_per_gpu": n_local, "entries_per_gpu": E, "dim": D,
                       "parallelism": "row-block shard x%d + all-gather(X) in %d column panel(s), panel-major persistent buffers"
                                      % (world, case["n_gather_panels"] if world > 1 else 1),
                       "frontend": args.frontend, "plan": (not args.no_plan), "rule": args.rule, "sparse_tasks": header.n_tasks,
                       "dense_windows": header.n_dense, "split_rows": header.n_split_rows,
                       "preprocess_ms": case["prep_warm_ms"], "preprocess_ms_cold": case["prep_cold_ms"],
                       "kernel_src_sha": kernel_src_sha()},
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
        if world == 1 and not args.no_sweep:
            out["sweep"] = sweep(fe, dev, args, rp, col)
        if world == 1 and not args.no_cpu_baseline:
            Xh = torch.randn(n_local * vworld, D, generator=torch.Generator().manual_seed(1234)).numpy()
            out["cpu_baseline"] = cpu_baseline(rp, col, Xh, D, n_local * vworld)
        print(json.dumps(out))
    if own_cache:
        shutil.rmtree(cache_dir, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def sweep(fe, dev, args, rp_head, col_head):
    """The other BASELINE points, timed in this process after the headline (fewer steps each)."""
    import torch
    entries = []
    steps, warmup = max(10, min(args.steps, 50)), 5
    plan = [("reddit", 32), ("reddit", 256), ("cora", 32), ("products_share", 256), ("c5_share", 128)]
    for wl, D in plan:
        if wl == args.workload and D == args.dim:
            continue
        n_local, e_local, _, vw, desc = WORKLOADS[wl]
        t0 = time.perf_counter()
        try:
            if wl == args.workload:
                rp, col = rp_head, col_head
            else:
                rp, col = make_local_block(wl, n_local, e_local, vw, 0)
            gen_s = time.perf_counter() - t0
            case = run_case(fe, dev, wl, D, rp, col, n_local, 1, 0, vw, steps, warmup, "f32", 0)
            rec = recorded_profile("%s_d%d" % (wl, D))
            traffic = rec.get("traffic_bytes") if rec else None
            src = None
            if traffic:
                src = "recorded: %s%s" % (rec.get("source"), " (STALE: kernel sources changed since)" if rec.get("stale") else "")
            roof = roofline_of(case, traffic, src)
            h = case["header"]
            e = {"workload": wl, "dim": D, "nodes": n_local, "entries": case["E"], "x_rows_resident": case["x_rows"],
                 "steps": steps, "ms_per_step": case["elapsed"] / steps * 1e3, "kernel_ms": case["kernel_ms"],
                 "value": case["E"] * D / (case["elapsed"] / steps), "unit": "edge*dim/s",
                 "sparse_tasks": h.n_tasks, "dense_windows": h.n_dense, "nnz_dense": h.nnz_dense, "split_rows": h.n_split_rows,
                 "preprocess_ms": case["prep_warm_ms"], "graph_gen_s": round(gen_s, 1), "roofline": roof, "desc": desc}
            if case.get("hip_graph_ms_per_step") is not None:
                e["hip_graph_ms_per_step"] = case["hip_graph_ms_per_step"]
                e["hip_graph_note"] = "the same step captured in a HIP graph and replayed: launch-bound workload, the Python call costs more than the kernel"
            if rec:
                for k in ("l2_hit_rate", "mfma_util_percent", "mfma_flops_per_launch"):
                    if rec.get(k) is not None:
                        e[k] = rec[k]
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
