"""Row-block sharding of the hybrid SpMM over the GPUs of one node (new design --
the reference is single-GPU, HC-SpMM_main.py:47-49,113).

Rank p owns a contiguous, window-aligned, nnz-balanced block of rows: A[rows_p, :] (columns stay
global), X[rows_p, :] and Z[rows_p, :].  One exchange step per SpMM: an all-gather of the X row
blocks (RCCL over xGMI when the backend is "nccl"); no reduction is needed because the output is
row-sharded.  Blocks are padded to a common height so the gather is a single
all_gather_into_tensor; global column ids are remapped once, on the host, to rows of the padded
gathered matrix.  The local product is whatever callable the caller supplies -- the HIP operator
in production (hcspmm.forward); CPU tests inject the oracle -- so this module contains no compute.
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
    """Z_local = A[rows_p, :] @ all_gather(X_local).

    local_spmm(X_full[P*pad_rows, D]) -> Z_local[n_local, D] is the local operator.
    With local_spmm_into(X_panel_full[P*pad_rows, w], Z_view[n_local, w]) and n_panels > 1 the feature
    columns are cut into n_panels panels: every panel's all-gather is enqueued up front
    (async_op=True, so RCCL runs them back to back on its own stream) and the product of panel k is
    issued as soon as gather k has landed -- it overlaps gather k+1, which is the longer of the two on
    xGMI.  Each gathered panel is a contiguous [P*pad_rows, w] matrix, which is
    also the layout the gather kernel likes best (one cache line per row for w = 32), and each product
    is written straight into its column slice of Z (strided operator, no concatenation).
    """

    def __init__(self, graph, local_spmm, group=None, local_spmm_into=None, n_panels=1):
        self.g = graph
        self.local_spmm = local_spmm
        self.local_spmm_into = local_spmm_into
        self.n_panels = n_panels if local_spmm_into is not None else 1
        self.group = group

    def _pad(self, X_local):
        g = self.g
        if X_local.shape[0] == g.pad_rows:
            return X_local
        pad = torch.zeros((g.pad_rows, X_local.shape[1]), dtype=X_local.dtype, device=X_local.device)
        pad[:g.n_local] = X_local
        return pad

    def _all_gather(self, full, part, async_op=False):
        if part.is_cuda and dist.get_backend(self.group) != "nccl":
            # rehearsal mode (e.g. several ranks sharing one GPU over gloo): stage through the host
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, part.cpu(), group=self.group)
            full.copy_(host)
            return None
        return dist.all_gather_into_tensor(full, part, group=self.group, async_op=async_op)

    def gather(self, X_local):
        g = self.g
        if g.world_size == 1:
            return X_local
        X_local = self._pad(X_local).contiguous()
        full = torch.empty((g.world_size * g.pad_rows, X_local.shape[1]), dtype=X_local.dtype, device=X_local.device)
        self._all_gather(full, X_local)
        return full

    def forward(self, X_local):
        g = self.g
        D = X_local.shape[1]
        if g.world_size == 1 or self.n_panels <= 1 or D % self.n_panels != 0:
            return self.local_spmm(self.gather(X_local))
        w = D // self.n_panels
        Xp = self._pad(X_local)
        fulls, works = [], []
        for p in range(self.n_panels):
            part = Xp[:, p * w:(p + 1) * w].contiguous()
            full = torch.empty((g.world_size * g.pad_rows, w), dtype=Xp.dtype, device=Xp.device)
            works.append(self._all_gather(full, part, async_op=True))
            fulls.append(full)
        Z = torch.empty((g.n_local, D), dtype=Xp.dtype, device=Xp.device)
        for p in range(self.n_panels):
            if works[p] is not None:
                works[p].wait()  # the current stream waits for gather p; gathers p+1.. keep running
            self.local_spmm_into(fulls[p], Z[:, p * w:(p + 1) * w])
        return Z

    __call__ = forward
