"""Row-block sharding of the hybrid SpMM over the GPUs of one node (new design --
the reference is single-GPU, HC-SpMM_main.py:47-49,113).

Rank p owns a contiguous, window-aligned, nnz-balanced block of rows: A[rows_p, :] (columns stay
global), X[rows_p, :] and Z[rows_p, :].  One exchange step per SpMM: an all-gather of the X row
blocks (RCCL over xGMI when the backend is "nccl"); no reduction is needed because the output is
row-sharded.  Blocks are padded to a common height so the gather is a single
all_gather_into_tensor; global column ids are remapped once, on the host, to rows of the padded
gathered matrix.  The local product is whatever callable the caller supplies -- the HIP operator
in production (hcspmm.forward_rect / forward_into); CPU tests inject the oracle -- so this module
contains no compute.

Data layout (what makes a step allocation- and copy-free): features live PANEL-MAJOR on every rank --
X_pm[n_panels][pad_rows][w], w = D / n_panels columns per panel (one 128-byte line per row for w = 32
fp32) -- and so does the result, Z_pm[n_panels][n_local][w].  Panel p of X is then a contiguous
[pad_rows, w] matrix that all_gather_into_tensor takes as it is (no .contiguous() copy), the gathered
panel is a contiguous [P*pad_rows, w] matrix -- the layout the gather kernel likes best -- and the
product of panel p is written straight into Z_pm[p].  Z_pm has the layout of X_pm, so the output of
one aggregation is the input of the next without a transpose.  All buffers (X_pm, the P-times-larger
gather targets, Z_pm, the split-row workspace) are allocated once by bind() and reused by every step.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_rows(row_pointers, world_size, align=16):
    """Contiguous row ranges [(r0, r1)] * world_size, boundaries multiples of `align` (row windows
    are never cut), balanced by stored entries + rows."""
    rp = np.asarray(row_pointers, dtype=np.int64)
    N = len(rp) - 1
    W = (N + align - 1) // align
    wstart = np.minimum(np.arange(W + 1) * align, N)
    cost = rp[wstart] + wstart  # entries + rows before each window boundary
    total = cost[-1]
    cuts = [0]
    for p in range(1, world_size):
        w = int(np.searchsorted(cost, total * p / world_size, side="left"))
        cuts.append(max(cuts[-1], min(w, W)))
    cuts.append(W)
    return [(int(wstart[cuts[p]]), int(wstart[cuts[p + 1]])) for p in range(world_size)]


class ShardedGraph:
    """The local row block of a CSR graph plus the column remap into the padded gathered X."""

    def __init__(self, row_pointers, column_index, ranges, rank):
        rp = np.asarray(row_pointers, dtype=np.int64)
        col = np.asarray(column_index, dtype=np.int64)
        self.ranges = list(ranges)
        self.rank = rank
        self.world_size = len(self.ranges)
        self.r0, self.r1 = self.ranges[rank]
        self.n_local = self.r1 - self.r0
        self.pad_rows = max(r1 - r0 for r0, r1 in self.ranges)
        e0, e1 = rp[self.r0], rp[self.r1]
        self.row_pointers = (rp[self.r0:self.r1 + 1] - e0).astype(np.int32)
        self.column_index = self.remap_columns(col[e0:e1]).astype(np.int32)

    @property
    def num_columns(self):
        """Rows of the padded gathered matrix = the column space of the local block (preprocess num_columns=)."""
        return self.world_size * self.pad_rows

    def remap_columns(self, cols):
        """global vertex id -> row of the padded gathered matrix (owner * pad_rows + local index)."""
        starts = np.array([r[0] for r in self.ranges], dtype=np.int64)
        owner = np.searchsorted(starts, cols, side="right") - 1
        return owner * self.pad_rows + (cols - starts[owner])

    @classmethod
    def from_local_block(cls, row_pointers_local, column_index_global, n_local, world_size, rank):
        """Equal-height blocks (weak-scaling benchmarks): every rank holds n_local rows, columns are
        global ids in [0, world_size * n_local); no padding, no remap needed."""
        g = cls.__new__(cls)
        g.ranges = [(p * n_local, (p + 1) * n_local) for p in range(world_size)]
        g.rank, g.world_size = rank, world_size
        g.r0, g.r1 = g.ranges[rank]
        g.n_local = g.pad_rows = n_local
        g.row_pointers = np.asarray(row_pointers_local, dtype=np.int32)
        g.column_index = np.asarray(column_index_global, dtype=np.int32)
        return g


class ShardedSpMM:
    """Z_local = A[rows_p, :] @ all_gather(X_local), panel by panel, with persistent buffers.

    local_spmm_into(X_panel_full[P*pad_rows, w], Z_view[n_local, w], workspace) is the local operator on one
    column panel (hcspmm.forward_into in production).  With n_panels > 1 every panel's all-gather is enqueued
    up front (async_op=True, so RCCL runs them back to back on its own stream) and the product of panel k is
    issued as soon as gather k has landed -- it overlaps gather k+1, which is the longer of the two on xGMI
    (DESIGN.md section 6).

    Use:  op.bind(D, dtype, device)            once: allocates X_pm, the gather targets, Z_pm, the workspace
          op.features()[...] = ...             fill the panel-major local features (or op.load_features(X))
          Z_pm = op.step()                     one SpMM: gathers + products; returns the persistent Z_pm
    forward(X_local) is the row-major convenience form (one strided copy in, one out); the step itself
    allocates and copies nothing.
    """

    def __init__(self, graph, local_spmm_into, group=None, n_panels=1, workspace_bytes=None, always_gather=False):
        """always_gather: take the collective path even in a one-rank group (a one-GPU box can then run the gather /
        product ordering through real RCCL; a one-rank job otherwise multiplies X_pm in place)."""
        self.always_gather = bool(always_gather)
        self.g = graph
        self.local_spmm_into = local_spmm_into
        self.n_panels = max(1, int(n_panels))
        self.group = group
        self._workspace_bytes = workspace_bytes  # callable(panel_width) -> bytes, or None
        self.D = None

    # ------------------------------------------------------------------ buffers
    def bind(self, D, dtype, device):
        g = self.g
        if D % self.n_panels != 0:
            raise ValueError("embedding_dim %d is not a multiple of n_panels %d" % (D, self.n_panels))
        self.D, self.w = int(D), int(D) // self.n_panels
        self.dtype, self.device = dtype, self._canonical(device)
        kw = dict(dtype=dtype, device=self.device)
        self.X_pm = torch.zeros((self.n_panels, g.pad_rows, self.w), **kw)  # padding rows stay zero
        self.Z_pm = torch.empty((self.n_panels, g.n_local, self.w), **kw)
        # world 1: the product reads X_pm itself
        self.gathered = [torch.empty((g.world_size * g.pad_rows, self.w), **kw) for _ in range(self.n_panels)] \
            if (g.world_size > 1 or self.always_gather) else None
        nbytes = int(self._workspace_bytes(self.w)) if self._workspace_bytes is not None else 0
        self.workspace = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=self.device) if nbytes else None
        self._host_stage = None
        return self

    @staticmethod
    def _canonical(device):
        """torch.device("cuda") and torch.device("cuda:0") name the same GPU: compare with the index filled in."""
        d = torch.device(device)
        if d.type == "cuda" and d.index is None:
            d = torch.device("cuda", torch.cuda.current_device())
        return d

    def features(self):
        """Panel-major view [n_panels, n_local, w] of this rank's features, to be filled in place."""
        return self.X_pm[:, :self.g.n_local, :]

    def load_features(self, X_local):
        """Row-major [n_local, D] -> the panel-major buffer (one strided device copy, no allocation)."""
        self.features().copy_(X_local.reshape(self.g.n_local, self.n_panels, self.w).permute(1, 0, 2))

    def buffers(self):
        """Every tensor a step touches, for allocation-freeness checks."""
        return [self.X_pm, self.Z_pm] + (self.gathered or []) + ([self.workspace] if self.workspace is not None else [])

    # ------------------------------------------------------------------ one SpMM
    def _all_gather(self, full, part, async_op):
        if part.is_cuda and dist.get_backend(self.group) != "nccl":
            # rehearsal mode (several ranks sharing one GPU over gloo): stage through the host
            if self._host_stage is None:
                self._host_stage = (torch.empty(full.shape, dtype=full.dtype), torch.empty(part.shape, dtype=part.dtype))
            host_full, host_part = self._host_stage
            host_part.copy_(part)
            dist.all_gather_into_tensor(host_full, host_part, group=self.group)
            full.copy_(host_full)
            return None
        return dist.all_gather_into_tensor(full, part, group=self.group, async_op=async_op)

    def step(self):
        g = self.g
        if g.world_size == 1 and not self.always_gather:
            for p in range(self.n_panels):
                self.local_spmm_into(self.X_pm[p], self.Z_pm[p], self.workspace)
            return self.Z_pm
        works = [self._all_gather(self.gathered[p], self.X_pm[p], async_op=True) for p in range(self.n_panels)]
        for p in range(self.n_panels):
            if works[p] is not None:
                works[p].wait()  # the current stream waits for gather p; gathers p+1.. keep running
            self.local_spmm_into(self.gathered[p], self.Z_pm[p], self.workspace)
        return self.Z_pm

    def forward(self, X_local):
        """Row-major convenience form: [n_local, D] -> a NEW [n_local, D] tensor (step() is the aliasing API: it returns the
        persistent Z_pm, which the next step overwrites)."""
        if self.D is None or self.D != X_local.shape[1] or self.dtype != X_local.dtype or self.device != self._canonical(X_local.device):
            self.bind(X_local.shape[1], X_local.dtype, X_local.device)
        self.load_features(X_local)
        Z_pm = self.step()
        if self.n_panels == 1:
            return Z_pm[0].clone()
        return Z_pm.permute(1, 0, 2).reshape(self.g.n_local, self.D)

    __call__ = forward
