"""hcspmm -- Python host glue over the C ABI of libhcspmm.so (include/hcspmm.h).

Mirrors the operator surface of the reference's `HCSPMM` extension module
(/root/reference/hybrid_kernel/hybrid_all.cpp:500-525): `preprocess`, `forward`,
`forward_more`, `forward_fixed32`, `forward_fixed64`, the fused variants and the `backward*`
aliases, with the same positional arguments and return lists, so code written against the
reference calls it unchanged.  Tensors are plumbing only (device memory + streams); all work
happens behind the C ABI.  There is NO CPU fallback: if libhcspmm.so is missing, or a feature
tensor is not on the GPU, the call raises.
"""
import collections
import ctypes
import os
import threading
import weakref

import torch

from .capi import (EPLAN, Header, PlanParams, RULE_AS_SHIPPED, RULE_INTENDED, RULE_INTENDED_GUARD, RULE_MI355X,
                   RULE_MI355X_WIDE, check, lib)

__all__ = [
    "preprocess", "forward", "forward_more", "forward_fixed32", "forward_fixed64", "forward_fixed32_fused",
    "forward_fixed64_fused", "forward_final_fused", "forward_final_fused_64", "forward_GIN_final_fused", "backward",
    "backward_fixed32", "backward_fixed32_fused", "backward_final_fused", "backward_fixed64",
    "backward_fixed64_fused", "backward_final_fused_64", "backward_GIN_final_fused", "loi_reorder",
    "apply_permutation", "weight_grad", "update", "plan_header", "forward_rect", "forward_into", "wide_threshold", "workspace_bytes", "fused_in_launch", "build_plan", "set_default_rule", "default_rule", "RULE_INTENDED", "RULE_INTENDED_GUARD",
    "RULE_AS_SHIPPED", "RULE_MI355X", "RULE_MI355X_WIDE", "mi355x_rule", "tune_plan",
]

# The window classifier preprocess() uses when the caller names none.  The reference's coefficients (rule 0) were fitted on an
# RTX 3090 and its paper says they hold only "while the GPU architecture ... remain[s] unchanged" (p.7): the default here is the
# refit made on MI355X with the reference's own procedure -- the NARROW one (RULE_MI355X), because preprocess is not told the
# embedding width and that set is never slower than rule 0 at any width (profiles/r04/ab_classifier_rules.log: -1 ... +25 %),
# while the wide set loses up to 15 % below 64 columns.  preprocess(rule="mi355x", dim=D) picks the set for a known width;
# HCSPMM_RULE=0 / set_default_rule(RULE_INTENDED) / rule=0 give the reference's hybrid_type bit for bit.
_DEFAULT_RULE = int(os.environ.get("HCSPMM_RULE", RULE_MI355X))
_PLAN_PARAMS = PlanParams(int(os.environ.get("HCSPMM_SPLIT_THRESHOLD", 0)), int(os.environ.get("HCSPMM_SEGMENT_LEN", 0)),
                          int(os.environ.get("HCSPMM_FUSE_IN_LAUNCH", 0)))


def set_default_rule(rule):
    """Classifier rule used by preprocess(): RULE_MI355X (default: the width-agnostic MI355X refit), RULE_MI355X_WIDE
    (mi355x_rule(D) picks between the two), or the reference's RULE_INTENDED, _GUARD, _AS_SHIPPED."""
    global _DEFAULT_RULE
    _DEFAULT_RULE = int(rule)


def default_rule():
    return _DEFAULT_RULE


def mi355x_rule(embedding_dim):
    """The MI355X refit of the window classifier for this embedding width (hcspmm.h: the boundary between
    the two sub-paths moves with D, and preprocess -- like the reference's -- is not told D)."""
    return RULE_MI355X if int(embedding_dim) < 64 else RULE_MI355X_WIDE


# ---------------------------------------------------------------------------------------------
# plan registry: data_ptr of a plan tensor -> host copy of its header.  The registry does NOT keep the
# tensor alive (a caller that preprocesses many graphs must be able to free their plans): an entry holds a
# weak reference and is valid only while that tensor lives -- while it does, its memory cannot be handed to
# another tensor, so the address is an unambiguous key; a dead entry is dropped and the header re-read (one
# 256-byte device read).  Each entry also remembers which (row_pointers, column_index) tensors the plan has
# been checked against: preprocess / build_plan register the pair the plan was built from; any other pair is
# fingerprinted on the device once (hcspmm_graph_fingerprint_device) and refused unless it matches the header.
# ---------------------------------------------------------------------------------------------
class _Entry:
    __slots__ = ("ref", "header", "graphs", "checked")

    def __init__(self, plan_t, header):
        self.ref = weakref.ref(plan_t)
        self.header = header
        self.graphs = collections.OrderedDict()  # (rowptr ptr, col ptr) -> (weakref, weakref)
        self.checked = None                      # (N, E, numel) hcspmm_plan_check has passed for


_REG = collections.OrderedDict()
_REG_LOCK = threading.Lock()
_REG_MAX = 256
_GRAPHS_MAX = 8


def _purge_locked():
    for k in [k for k, e in _REG.items() if e.ref() is None]:
        del _REG[k]
    while len(_REG) >= _REG_MAX:  # least recently used first
        _REG.popitem(last=False)


def _register(plan_t, header, row_pointers=None, column_index=None):
    e = _Entry(plan_t, header)
    if row_pointers is not None and column_index is not None and row_pointers.is_cuda:
        e.graphs[(row_pointers.data_ptr(), column_index.data_ptr())] = (weakref.ref(row_pointers), weakref.ref(column_index))
    with _REG_LOCK:
        _purge_locked()
        _REG[plan_t.data_ptr()] = e
    return e


def _entry(row_nzr, num_nodes=None, num_edges=None):
    if row_nzr is None or row_nzr.numel() < Header.WORDS or row_nzr.dtype != torch.int32:
        return None
    key = row_nzr.data_ptr()
    with _REG_LOCK:
        e = _REG.get(key)
        if e is not None:
            if e.ref() is not None:
                _REG.move_to_end(key)
                return e
            del _REG[key]
    host = row_nzr[:Header.WORDS].cpu().contiguous()
    h = Header.from_buffer_copy(host.numpy().tobytes())
    if h.magic != Header.MAGIC:
        return None
    if num_nodes is not None and check(lib().hcspmm_plan_check(ctypes.byref(h), num_nodes, num_edges, row_nzr.numel()),
                                       soft=True) != 0:
        return None
    return _register(row_nzr, h)


def plan_header(row_nzr, num_nodes=None, num_edges=None):
    """Host copy of the plan header carried by `row_nzr`, or None for the reference's [0]
    placeholder.  A plan tensor first seen here (e.g. a clone) costs one small device read."""
    e = _entry(row_nzr, num_nodes, num_edges)
    return e.header if e is not None else None


def _verify_graph(e, row_pointers, column_index):
    """The plan of entry `e` was built for ONE graph; N and E alone do not identify it (a LOI reorder keeps both)."""
    key = (row_pointers.data_ptr(), column_index.data_ptr())
    hit = e.graphs.get(key)
    if hit is not None and hit[0]() is not None and hit[1]() is not None:
        return
    L = lib()
    out = torch.empty(1, dtype=torch.int64, device=row_pointers.device)
    stream = torch.cuda.current_stream(row_pointers.device)
    with torch.cuda.device(row_pointers.device):
        check(L.hcspmm_graph_fingerprint_device(_ptr(row_pointers), _ptr(column_index), row_pointers.numel() - 1,
                                                column_index.numel(), _ptr(out), ctypes.c_void_p(stream.cuda_stream)))
    got = int(out.item()) & 0xFFFFFFFFFFFFFFFF
    if got != e.header.fingerprint:
        raise RuntimeError("hcspmm: plan does not match this graph (row_pointers / column_index differ from the ones "
                           "the plan was built from) [code %d]" % EPLAN)
    while len(e.graphs) >= _GRAPHS_MAX:
        e.graphs.popitem(last=False)
    e.graphs[key] = (weakref.ref(row_pointers), weakref.ref(column_index))


def wide_threshold(row_nzr, embedding_dim, dtype=torch.float32):
    """Rows of the sparse path with more entries than this are summed by a whole wave (shuffle-tree
    combine) instead of one lane group in CSR order; see hcspmm_wide_threshold in include/hcspmm.h."""
    h = plan_header(row_nzr)
    return int(lib().hcspmm_wide_threshold_typed(ctypes.byref(h) if h is not None else None, int(embedding_dim),
                                                 _DTYPES[dtype]))


def workspace_bytes(row_nzr, embedding_dim):
    """Bytes of fp32 workspace a forward with this plan and width needs (partial sums of split rows); 0 without a plan."""
    h = plan_header(row_nzr)
    return int(lib().hcspmm_workspace_bytes(ctypes.byref(h), int(embedding_dim))) if h is not None else 0


def fused_in_launch(row_nzr, embedding_dim, hidden_dim):
    """Form forward_*_fused takes with this plan and shape (hcspmm_fused_in_launch, include/hcspmm.h): 0 = two launches,
    1 = dense-tile windows update inside the hybrid launch, 2 = the sparse-row path as well (row-tile form)."""
    h = plan_header(row_nzr)
    return int(lib().hcspmm_fused_in_launch(ctypes.byref(h), int(embedding_dim), int(hidden_dim))) if h is not None else 0


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else 0)


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already current (the context manager costs a few microseconds,
    which is the whole kernel time on a Cora-scale graph)."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        self.ctx = None if device.index is None or device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _i32_host(t):
    t = t.detach()
    if t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() > (1 << 16):
        # large device arrays come back through a pinned buffer: a pageable D2H copy runs at ~1.4 GB/s
        # (first-touch page faults inside the copy), a pinned one at PCIe rate
        host = torch.empty(t.shape, dtype=torch.int32, pin_memory=True)
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host
    return t.to(device="cpu", dtype=torch.int32).contiguous()


def preprocess(column_index, row_pointers, num_nodes, num_edges, num_row_windows, rule=None, dim=None, num_columns=None):
    """HCSPMM.preprocess (hybrid_all.cpp:13-17,501; hybrid_all_kernel.cu:339-408).

    Argument order as the reference: column_index FIRST (HC-SpMM_main.py:52).  `num_edges` is
    ignored in favour of column_index.size(0).  Runs on the host (north_star),
    returns [blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr] on the device
    of `column_index`; `row_nzr` carries the MI355X launch plan, `col_nzr` stays the [0] placeholder.
    `rule` (not in the reference): a HCSPMM_RULE_* number, or "mi355x" together with `dim` = the embedding
    width the graph will be multiplied at (picks the narrow or the wide MI355X refit).
    `num_columns` (not in the reference, whose graphs are square): rows of the matrix the column ids index, for
    a row block of a sharded graph.  Column ids outside [0, num_columns) raise -- on the GPU they would be
    out-of-bounds gathers.
    """
    L = lib()
    dev = column_index.device
    col_h = _i32_host(column_index)
    rp_h = _i32_host(row_pointers)
    N = int(rp_h.numel()) - 1
    E = int(col_h.numel())
    if int(num_nodes) != N:
        raise RuntimeError("preprocess: num_nodes (%d) != row_pointers.size(0)-1 (%d)" % (num_nodes, N))
    W = (N + 15) // 16
    if int(num_row_windows) != W:
        raise RuntimeError("preprocess: num_row_windows (%d) != ceil(N/16) (%d)" % (num_row_windows, W))
    on_gpu = dev.type == "cuda"
    # host outputs in pinned memory when they are headed for the GPU (torch's caching host allocator recycles the blocks
    # from call to call): the uploads run at PCIe rate and asynchronously instead of through a pageable staging copy
    bp = torch.empty(W, dtype=torch.int32, pin_memory=on_gpu)
    ht = torch.empty(W, dtype=torch.int32, pin_memory=on_gpu)
    e2c = torch.empty(E, dtype=torch.int32, pin_memory=on_gpu)
    # edgeToRow is the plain CSR row expansion: made on the device when the graph lives there -- fill_edgeToRow
    # (K.cu:314-337) as one small HIP kernel of the library, enqueued now so that it runs under the host passes below
    e2r = None
    if on_gpu:
        e2r = torch.empty(E, dtype=torch.int32, device=dev)
        rp_dev = row_pointers.to(device=dev, dtype=torch.int32).contiguous()  # (no copy when it already is)
        with torch.cuda.device(dev):
            check(L.hcspmm_edge_to_row_device(_ptr(rp_dev), N, E, _ptr(e2r),
                                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    e2r_h = None if on_gpu else torch.empty(E, dtype=torch.int32)
    if rule == "mi355x":
        if dim is None:
            raise RuntimeError('preprocess: rule="mi355x" needs dim= (the embedding width)')
        rule = mi355x_rule(dim)
    r = _DEFAULT_RULE if rule is None else int(rule)
    M = N if num_columns is None else int(num_columns)
    check(L.hcspmm_preprocess_host(_ptr(rp_h), _ptr(col_h), N, E, M, r, 0, _ptr(bp), _ptr(e2c), _ptr(e2r_h), _ptr(ht)))
    if not on_gpu:
        e2r = e2r_h
    # the window products go up while the host builds the plan from them (pinned memory: asynchronous PCIe-rate copies)
    up = [t.to(dev, non_blocking=True) for t in (bp, e2c, e2r, ht)]  # .to() is a no-op for the device-made e2r
    words = ctypes.c_int64(0)
    check(L.hcspmm_plan_words(_ptr(rp_h), N, E, _ptr(bp), _ptr(ht), ctypes.byref(_PLAN_PARAMS), ctypes.byref(words)))
    plan = torch.empty(max(int(words.value), Header.WORDS), dtype=torch.int32, pin_memory=on_gpu)
    check(L.hcspmm_plan_build(_ptr(rp_h), _ptr(col_h), N, E, M, _ptr(bp), _ptr(e2c), _ptr(ht),
                              ctypes.byref(_PLAN_PARAMS), _ptr(plan), plan.numel()))
    h = Header.from_buffer_copy(plan[:Header.WORDS].numpy().tobytes())
    plan = plan[:h.total_words]  # hcspmm_plan_words sizes for the larger of the two layouts (column slices or none)
    outs = up + [plan.to(dev, non_blocking=True)]
    _register(outs[4], h, row_pointers, column_index)
    col_nzr = torch.zeros(1, dtype=torch.int32, device=dev)
    return [outs[0], outs[1], outs[2], outs[3], outs[4], col_nzr]


def build_plan(row_pointers, column_index, blockPartition, edgeToColumn, hybrid_type, device=None,
               split_threshold=0, segment_len=0, num_columns=None, fuse_in_launch=False, slice_threshold=0, n_slices=0,
               panel_cols=0):
    """Launch plan for an arbitrary window classification (e.g. every window forced onto one sub-path,
    or a classifier of the caller's own): -> plan tensor to pass as `row_nzr`.  fuse_in_launch: 1 (or True) = the fused
    operators update this plan's dense-tile windows inside the hybrid launch, 2 = the sparse-row path as well (row-tile
    form; include/hcspmm.h hcspmm_forward_fused).
    slice_threshold / n_slices: XCD-affine column slices (hcspmm_plan_params; 0 = automatic, < 0 = off).
    panel_cols: feature columns per pass of the sparse-row path (0 = chosen at launch, < 0 = one pass; see tune_plan)."""
    L = lib()
    rp_h, col_h = _i32_host(row_pointers), _i32_host(column_index)
    bp_h, e2c_h, ht_h = _i32_host(blockPartition), _i32_host(edgeToColumn), _i32_host(hybrid_type)
    N, E = rp_h.numel() - 1, col_h.numel()
    params = PlanParams(int(split_threshold), int(segment_len), int(fuse_in_launch), int(slice_threshold), int(n_slices),
                        int(panel_cols)) \
        if (split_threshold or segment_len or fuse_in_launch or slice_threshold or n_slices or panel_cols) else _PLAN_PARAMS
    words = ctypes.c_int64(0)
    check(L.hcspmm_plan_words(_ptr(rp_h), N, E, _ptr(bp_h), _ptr(ht_h), ctypes.byref(params), ctypes.byref(words)))
    plan = torch.empty(max(int(words.value), Header.WORDS), dtype=torch.int32)
    check(L.hcspmm_plan_build(_ptr(rp_h), _ptr(col_h), N, E, N if num_columns is None else int(num_columns), _ptr(bp_h),
                              _ptr(e2c_h), _ptr(ht_h), ctypes.byref(params), _ptr(plan), plan.numel()))
    h = Header.from_buffer_copy(plan[:Header.WORDS].numpy().tobytes())
    plan = plan[:h.total_words]
    dev = torch.device(device) if device is not None else row_pointers.device
    plan_d = plan.to(dev)
    _register(plan_d, h, row_pointers, column_index)
    return plan_d


def tune_plan(row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, embedding_dim,
              dtype=torch.float32, num_columns=None, candidates=None, steps=20, fuse_in_launch=False):
    """Measure, on this GPU and THIS graph, the plan variants whose best choice no size rule predicts -- column slices on /
    off and the panel width of the sparse-row path -- and return (best plan tensor, report).  The automatic choices are tuned
    on power-law graphs from 10 K to 16 M nodes; between them (e.g. 66 K-130 K nodes) another panel width can be 5-17 %
    faster (profiles/r03/ab_panel_midsize.log, ab_slices_midsize.log).  Costs a handful of plan builds and
    len(candidates) * (3 + steps) launches: worth it for a graph that will be multiplied thousands of times (training).
    candidates: list of dicts of build_plan keywords (slice_threshold, n_slices, panel_cols, ...); default = slices
    {automatic, off, 256 x 8} x panels {automatic, 32, 64, one pass} (panels only for embedding_dim >= 64).  The results of
    all variants agree within the 1e-5 bar; rows short enough to be summed in CSR order by every variant agree bit for bit."""
    dev = row_pointers.device
    if not row_pointers.is_cuda:
        raise RuntimeError("tune_plan measures on the GPU: graph tensors must be CUDA tensors")
    D = int(embedding_dim)
    N = row_pointers.size(0) - 1
    M = N if num_columns is None else int(num_columns)
    if candidates is None:
        panels = [0, 32, 64, -1] if D >= 64 else [0]  # (fp32 columns: 16-bit features take twice as many per pass)
        candidates = [dict(slice_threshold=s, panel_cols=p) for s in (0, -1, 256) for p in panels]
    X = torch.randn(M, D, device=dev).to(dtype)
    Z = torch.empty(N, D, dtype=dtype, device=dev)
    col_nzr = torch.zeros(1, dtype=torch.int32, device=dev)
    report, best, seen = [], None, set()
    for kw in candidates:
        plan = build_plan(row_pointers, column_index, blockPartition, edgeToColumn, hybrid_type, num_columns=M,
                          fuse_in_launch=fuse_in_launch, **kw)
        h = plan_header(plan)
        key = (h.n_slices, h.slice_threshold, h.panel_cols, h.n_slice_tasks)
        if key in seen:  # e.g. "slices off" on a graph the automatic rule leaves unsliced anyway
            continue
        seen.add(key)
        ws = torch.empty(max(workspace_bytes(plan, D) // 4, 1), dtype=torch.float32, device=dev)
        a = (row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, plan, col_nzr)
        for _ in range(3):
            forward_into(X, Z, *a, workspace=ws)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(steps):
            forward_into(X, Z, *a, workspace=ws)
        e.record()
        e.synchronize()
        ms = s.elapsed_time(e) / steps
        report.append(dict(kw, ms=ms, n_slices=h.n_slices))
        if best is None or ms < best[0]:
            best = (ms, plan)
    return best[1], sorted(report, key=lambda r: r["ms"])


def _check_input(t, name):
    # hybrid_all.cpp:185-187 CHECK_CUDA / CHECK_CONTIGUOUS, same messages
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


# feature element types of hcspmm_forward_typed (include/hcspmm.h HCSPMM_DTYPE_*)
_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _graph_args(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                rect=False, dtypes=(torch.float32,)):
    for t, n in ((X, "input"), (row_pointers, "nodePointer"), (column_index, "edgeList"),
                 (blockPartition, "blockPartition"), (edgeToColumn, "edgeToColumn"), (edgeToRow, "edgeToRow")):
        _check_input(t, n)
    if X.dtype not in dtypes or X.dim() != 2:
        raise RuntimeError("input must be a 2-D %s tensor" % " / ".join(str(d).replace("torch.", "") for d in dtypes))
    N = row_pointers.size(0) - 1
    E = column_index.size(0)
    D = X.size(1)
    if X.size(0) != N and not rect:
        raise RuntimeError("input has %d rows but the graph has %d nodes" % (X.size(0), N))
    h = _checked_header(row_nzr, row_pointers, column_index, N, E, X.size(0))
    return N, E, D, h


def _checked_header(row_nzr, row_pointers, column_index, N, E, x_rows):
    """Header of the plan in `row_nzr` (None: the reference's placeholder -> plan-free kernel), after checking
    that the plan belongs to THIS graph and that X has every row the plan gathers."""
    e = _entry(row_nzr, N, E) if (row_nzr is not None and row_nzr.is_cuda) else None
    if e is None:
        return None
    if e.checked != (N, E, row_nzr.numel()):
        check(lib().hcspmm_plan_check(ctypes.byref(e.header), N, E, row_nzr.numel()))
        e.checked = (N, E, row_nzr.numel())
    if x_rows < e.header.num_columns:
        raise RuntimeError("input has %d rows but the plan gathers from %d" % (x_rows, e.header.num_columns))
    _verify_graph(e, row_pointers, column_index)
    return e.header


def _spmm(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr,
          Z=None, rect=False):
    L = lib()
    N, E, D, h = _graph_args(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type,
                             row_nzr, rect, dtypes=tuple(_DTYPES))
    if Z is None:
        Z = torch.empty((N, D), dtype=X.dtype, device=X.device)
    ws, ws_bytes = None, 0
    if h is not None:
        ws_bytes = int(L.hcspmm_workspace_bytes(ctypes.byref(h), D))
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=X.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
    with _on_device(X.device):
        check(L.hcspmm_forward_typed(_ptr(X), X.size(0), D, _ptr(Z), D, _DTYPES[X.dtype], _ptr(row_pointers), _ptr(column_index),
                                     _ptr(blockPartition), _ptr(edgeToColumn), _ptr(edgeToRow), _ptr(hybrid_type),
                                     _ptr(row_nzr) if h is not None else ctypes.c_void_p(0),
                                     ctypes.byref(h) if h is not None else None, N, E, D, _ptr(ws), ws_bytes, stream))
    return Z


def forward(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr):
    """HCSPMM.forward -> [A*X]  (hybrid_all.cpp:194-221; any embedding_dim).  X may also be float16 / bfloat16
    (the paper's half-precision variants, Table VII): rows are gathered as stored, summed in fp32 in the fp32
    path's order and rounded once -- Z has X's dtype."""
    return [_spmm(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr)]


def forward_rect(X_full, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                 col_nzr):
    """Row-block form used by the multi-GPU shard (hcspmm.sharded): A is n_local x M with column ids
    indexing the rows of X_full (M x D, the all-gathered embedding matrix) -> [Z_local (n_local x D)]."""
    return [_spmm(X_full, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                  col_nzr, rect=True)]


def forward_into(X, Z, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                 col_nzr, workspace=None):
    """Strided form: Z[:, :] = A @ X where X and Z may be column slices of wider matrices (unit inner
    stride, any row stride) and X may have any number of rows (column ids index them).  Used by the
    multi-GPU shard to multiply one gathered column panel at a time straight into its slice of Z.
    `workspace`: optional caller-kept fp32 buffer of at least workspace_bytes(row_nzr, D) bytes."""
    L = lib()
    for t, n in ((row_pointers, "nodePointer"), (column_index, "edgeList"), (blockPartition, "blockPartition"),
                 (edgeToColumn, "edgeToColumn"), (edgeToRow, "edgeToRow")):
        _check_input(t, n)
    for t, n in ((X, "input"), (Z, "output")):
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % n)
        if t.dtype not in _DTYPES or t.dtype != X.dtype or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.size(1):
            raise RuntimeError("%s must be a 2-D float32 / float16 / bfloat16 view with unit inner stride" % n)
    N, E, D = row_pointers.size(0) - 1, column_index.size(0), X.size(1)
    if Z.size(0) != N or Z.size(1) != D:
        raise RuntimeError("output must be [num_nodes, embedding_dim]")
    h = _checked_header(row_nzr, row_pointers, column_index, N, E, X.size(0))
    ws, ws_bytes = None, 0
    if h is not None:
        ws_bytes = int(L.hcspmm_workspace_bytes(ctypes.byref(h), D))
        if ws_bytes:
            if workspace is not None and workspace.numel() * workspace.element_size() >= ws_bytes:
                ws = workspace  # a caller-kept buffer: nothing is allocated in the step (hcspmm.sharded)
            else:
                ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=X.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
    with _on_device(X.device):
        check(L.hcspmm_forward_typed(_ptr(X), X.size(0), X.stride(0), _ptr(Z), Z.stride(0), _DTYPES[X.dtype], _ptr(row_pointers),
                                     _ptr(column_index), _ptr(blockPartition), _ptr(edgeToColumn), _ptr(edgeToRow),
                                     _ptr(hybrid_type), _ptr(row_nzr) if h is not None else ctypes.c_void_p(0),
                                     ctypes.byref(h) if h is not None else None, N, E, D, _ptr(ws), ws_bytes, stream))
    return Z


def update(X, W):
    """X @ W for X [N, D] (fp32, contiguous) and W [D, H] (any strides: a transposed view needs no copy) through the
    library's streaming MFMA update kernel (hcspmm_dense_update) -- the layers' torch.mm(X, weights), which for N in the
    millions and D, H of a few dozen is a stream over X.  Returns None when the operands are not of that kind (caller: torch.mm)."""
    if not (X.is_cuda and W.is_cuda and X.dtype == W.dtype == torch.float32 and X.dim() == W.dim() == 2
            and X.size(1) == W.size(0) and X.is_contiguous() and X.size(0) > 0 and W.size(1) > 0):
        return None
    out = torch.empty((X.size(0), W.size(1)), dtype=torch.float32, device=X.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
    with _on_device(X.device):
        check(lib().hcspmm_dense_update(_ptr(X), _ptr(W), W.stride(0), W.stride(1), _ptr(out), X.size(0), X.size(1), W.size(1), stream))
    return out


def weight_grad(A, B):
    """dW = A^T B for A [N, D], B [N, H] (fp32, unit inner stride): the weight gradient of the update GEMM in the
    layers' backward passes (reference: torch.mm(X.t(), d_out), GNN_model.py:79,101,...) through the split-K MFMA
    kernel of hcspmm_weight_grad.  Returns None when the shape is outside the kernel's range (caller: library GEMM)."""
    L = lib()
    if not (A.is_cuda and B.is_cuda and A.dtype == B.dtype == torch.float32 and A.dim() == B.dim() == 2
            and A.size(0) == B.size(0) and A.stride(1) == 1 and B.stride(1) == 1):
        return None
    N, D, H = A.size(0), A.size(1), B.size(1)
    ws_bytes = int(L.hcspmm_weight_grad_workspace(N, D, H))
    if ws_bytes == 0:
        return None
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=A.device)
    out = torch.empty((D, H), dtype=torch.float32, device=A.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
    with torch.cuda.device(A.device):
        check(L.hcspmm_weight_grad(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), N, D, H, _ptr(ws), ws_bytes, stream))
    return out


# The reference's dim-specialised variants compute the same product (hybrid_all.cpp:223-308);
# forward_fixed32 silently truncated to 32 columns for D > 32 -- not reproduced.
forward_more = forward
forward_fixed32 = forward
forward_fixed64 = forward


def _fused(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr,
           weights, output=None):
    L = lib()
    N, E, D, h = _graph_args(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type,
                             row_nzr)
    if not weights.is_cuda or weights.dtype != torch.float32 or weights.dim() != 2 or weights.size(0) != D:
        raise RuntimeError("weights must be a CUDA float32 tensor of shape [embedding_dim, hidden_dim]")
    H = weights.size(1)
    if output is None:
        output = torch.empty((N, H), dtype=torch.float32, device=X.device)
    else:
        _check_input(output, "output")
        if output.dtype != torch.float32 or output.numel() != N * H:
            raise RuntimeError("output must be float32 with num_nodes*hidden_dim elements")
    out2 = torch.empty((N, D), dtype=torch.float32, device=X.device)
    ws, ws_bytes = None, 0
    if h is not None:
        ws_bytes = int(L.hcspmm_workspace_bytes(ctypes.byref(h), D))
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=X.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
    with _on_device(X.device):
        check(L.hcspmm_forward_fused(_ptr(X), _ptr(output), _ptr(out2), _ptr(weights), weights.stride(0),
                                     weights.stride(1), H, _ptr(row_pointers), _ptr(column_index),
                                     _ptr(blockPartition), _ptr(edgeToColumn), _ptr(edgeToRow), _ptr(hybrid_type),
                                     _ptr(row_nzr) if h is not None else ctypes.c_void_p(0),
                                     ctypes.byref(h) if h is not None else None, N, E, D, _ptr(ws), ws_bytes, stream))
    return [output, out2]


def forward_fixed32_fused(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                          col_nzr, weights):
    """-> [(A*X)*weights, A*X]  (hybrid_all.cpp:310-339).  `weights` may be a transposed view."""
    return _fused(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr,
                  weights)


forward_fixed64_fused = forward_fixed32_fused
forward_GIN_final_fused = forward_fixed32_fused


def forward_final_fused(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                        col_nzr, weights, output):
    """-> [output (the caller's tensor, written in place), A*X]  (hybrid_all.cpp:405-435)."""
    return _fused(X, row_pointers, column_index, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr, col_nzr,
                  weights, output)


forward_final_fused_64 = forward_final_fused

# hybrid_all.cpp:516-523: every backward* name is bound to the matching forward function.
backward = forward
backward_fixed32 = forward_fixed32
backward_fixed32_fused = forward_fixed32_fused
backward_final_fused = forward_final_fused
backward_fixed64 = forward_fixed64
backward_fixed64_fused = forward_fixed64_fused
backward_final_fused_64 = forward_final_fused_64
backward_GIN_final_fused = forward_GIN_final_fused


def loi_reorder(row_pointers, column_index, variant="new_direct", batch=0, list_cap=0, threads=0):
    """LOI layout reorder (LOI.cpp:660-805 + main's output order) -> (perm[N], group_sizes).
    variant "fast" = the relaxed parallel form (hcspmm_loi_reorder_fast: capped column-list walks, `batch` seeds per round,
    deterministic for a given (batch, list_cap) whatever `threads`; NOT the reference's permutation -- batch=1, list_cap=-1 is).
    variant "new" = reorder_plus_new (LOI.cpp:505-658, symmetric-graph form); "plus_direct" / "plus" = the windowed
    variants reorder_plus_direct / reorder_plus (LOI.cpp:286-484 / :98-284; defined for graphs of at least 50 rows
    without empty rows -- other inputs raise); a HCSPMM_LOI_* number is accepted as well."""
    L = lib()
    rp = _i32_host(row_pointers)
    col = _i32_host(column_index)
    N, E = rp.numel() - 1, col.numel()
    perm = torch.empty(N, dtype=torch.int32)
    gs = torch.empty(max(N, 1), dtype=torch.int32)
    ng = ctypes.c_int64(0)
    if variant == "fast":
        params = (ctypes.c_int32 * 4)(int(batch), int(list_cap), int(threads), 0)
        check(L.hcspmm_loi_reorder_fast(_ptr(rp), _ptr(col), N, E, ctypes.cast(params, ctypes.c_void_p), _ptr(perm), _ptr(gs),
                                        ctypes.byref(ng)))
        return perm, gs[:ng.value].clone()
    v = variant if isinstance(variant, int) else {"new_direct": 0, "new": 1, "plus_direct": 2, "plus": 3}[variant]
    check(L.hcspmm_loi_reorder_variant(_ptr(rp), _ptr(col), N, E, v, _ptr(perm), _ptr(gs), ctypes.byref(ng)))
    return perm, gs[:ng.value].clone()


def apply_permutation(row_pointers, column_index, perm):
    """Relabel a CSR graph with a LOI permutation -> (row_pointers', column_index')."""
    L = lib()
    rp = _i32_host(row_pointers)
    col = _i32_host(column_index)
    p = _i32_host(perm)
    N, E = rp.numel() - 1, col.numel()
    rp2 = torch.empty(N + 1, dtype=torch.int32)
    col2 = torch.empty(E, dtype=torch.int32)
    check(L.hcspmm_apply_permutation(_ptr(rp), _ptr(col), N, E, _ptr(p), _ptr(rp2), _ptr(col2)))
    return rp2, col2
