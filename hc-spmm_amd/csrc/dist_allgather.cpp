// dist_allgather.cpp -- libhcspmm_dist.so: the multi-GPU step of the row-block shard as a C ABI (include/hcspmm_dist.h).
// New (the reference is single-GPU, HC-SpMM_main.py:47-49); the same schedule as hcspmm/sharded.py ShardedSpMM.step():
// all panel gathers enqueued up front on the communication stream, product p on the compute stream behind gather p.
// RCCL over xGMI is point-to-point: an all-gather is bound by one link per peer whatever the panel width, so the panels
// exist for overlap (product k under gather k + 1) and for the kernel's own one-cache-line-per-row access, not for the wire.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <new>
#include <vector>

#include "hcspmm_dist.h"
#include "host_util.h"

namespace {
thread_local int g_last_error = 0;
int fail(int code) {
  g_last_error = code;
  return HCSPMM_EHIP;
}
inline size_t elem_bytes(int dtype) { return dtype == HCSPMM_DTYPE_F32 ? 4 : 2; }
inline ncclDataType_t nccl_type(int dtype) {
  return dtype == HCSPMM_DTYPE_F32 ? ncclFloat32 : (dtype == HCSPMM_DTYPE_F16 ? ncclFloat16 : ncclBfloat16);
}
}  // namespace

struct hcspmm_dist_ctx {
  std::vector<hipEvent_t> landed;  // gather p has landed
  hipEvent_t reached = nullptr;    // the compute stream has reached this step
};

extern "C" int hcspmm_dist_last_error(void) { return g_last_error; }

// hcspmm/sharded.py partition_rows, restated: cut where (entries + rows before a window boundary) crosses p / world of the total
extern "C" int hcspmm_dist_partition_rows(const int32_t* rowptr, int64_t N, int world, int64_t* ranges) {
  if (!rowptr || !ranges || N < 0 || world < 1) return HCSPMM_EINVAL;
  if (rowptr[N] < 0 || !hcspmm::csr_row_pointers_ok(rowptr, N, rowptr[N])) return HCSPMM_EINVAL;  // start at 0, never decrease
  const int64_t W = (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H;
  auto wstart = [&](int64_t w) { return std::min<int64_t>(w * HCSPMM_BLK_H, N); };
  auto cost = [&](int64_t w) { return (int64_t)rowptr[wstart(w)] + wstart(w); };
  const int64_t total = cost(W);
  int64_t prev = 0;
  for (int p = 0; p < world; ++p) {
    int64_t cutw = W;
    if (p + 1 < world) {
      // first window boundary whose cost reaches total * (p + 1) / world (compared without rounding: cost * world >= total * (p + 1))
      int64_t lo = 0, hi = W;
      while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if ((__int128)cost(mid) * world >= (__int128)total * (p + 1)) hi = mid;
        else lo = mid + 1;
      }
      cutw = std::max(prev, std::min(lo, W));
    }
    ranges[2 * p] = wstart(prev);
    ranges[2 * p + 1] = wstart(cutw);
    prev = cutw;
  }
  return HCSPMM_OK;
}

extern "C" int hcspmm_dist_extract_block(const int32_t* rowptr, const int32_t* col, int64_t N, int world, const int64_t* ranges,
                                         int rank, int32_t* rp_out, int32_t* col_out, int64_t* pad_rows_out) {
  if (!rowptr || !ranges || !rp_out || N < 0 || world < 1 || rank < 0 || rank >= world) return HCSPMM_EINVAL;
  // the interface has no entry count of its own: rowptr[N] is it, once the pointers are known to start at 0 and never to
  // decrease (every [rowptr[r], rowptr[r + 1]) then lies inside the caller's column_index[0, rowptr[N]))
  if (rowptr[N] < 0 || !hcspmm::csr_row_pointers_ok(rowptr, N, rowptr[N])) return HCSPMM_EINVAL;
  int64_t pad = 0;
  for (int p = 0; p < world; ++p) {
    if (ranges[2 * p] < 0 || ranges[2 * p + 1] < ranges[2 * p] || ranges[2 * p + 1] > N) return HCSPMM_EINVAL;
    if (p > 0 && ranges[2 * p] != ranges[2 * p - 1]) return HCSPMM_EINVAL;
    pad = std::max(pad, ranges[2 * p + 1] - ranges[2 * p]);
  }
  if (ranges[0] != 0 || ranges[2 * world - 1] != N) return HCSPMM_EINVAL;
  if (pad * world > INT32_MAX) return HCSPMM_ERANGE;
  const int64_t r0 = ranges[2 * rank], r1 = ranges[2 * rank + 1], e0 = rowptr[r0], e1 = rowptr[r1];
  if (e1 > e0 && (!col || !col_out)) return HCSPMM_EINVAL;
  for (int64_t r = r0; r <= r1; ++r) rp_out[r - r0] = (int32_t)(rowptr[r] - e0);
  std::vector<int64_t> starts((size_t)world);
  for (int p = 0; p < world; ++p) starts[(size_t)p] = ranges[2 * p];
  for (int64_t e = e0; e < e1; ++e) {
    const int64_t v = col[e];
    if (v < 0 || v >= N) return HCSPMM_EINVAL;
    // owner: the LAST rank whose first row is <= v (empty blocks share a start with their successor and own nothing)
    const int64_t owner = (std::upper_bound(starts.begin(), starts.end(), v) - starts.begin()) - 1;
    col_out[e - e0] = (int32_t)(owner * pad + (v - starts[(size_t)owner]));
  }
  if (pad_rows_out) *pad_rows_out = pad;
  return HCSPMM_OK;
}

extern "C" int hcspmm_dist_create(int max_panels, hcspmm_dist_ctx** out) {
  if (max_panels <= 0 || max_panels > 4096 || !out) return HCSPMM_EINVAL;
  hcspmm_dist_ctx* c = new (std::nothrow) hcspmm_dist_ctx;
  if (!c) return HCSPMM_ENOMEM;
  hipError_t e = hipEventCreateWithFlags(&c->reached, hipEventDisableTiming);
  for (int p = 0; e == hipSuccess && p < max_panels; ++p) {
    hipEvent_t ev = nullptr;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) c->landed.push_back(ev);
  }
  if (e != hipSuccess) {
    hcspmm_dist_destroy(c);
    return fail((int)e);
  }
  *out = c;
  return HCSPMM_OK;
}

extern "C" void hcspmm_dist_destroy(hcspmm_dist_ctx* c) {
  if (!c) return;
  for (hipEvent_t ev : c->landed) (void)hipEventDestroy(ev);
  if (c->reached) (void)hipEventDestroy(c->reached);
  delete c;
}

extern "C" int hcspmm_dist_step(hcspmm_dist_ctx* c, const hcspmm_dist_step_args* a) {
  if (!c || !a) return HCSPMM_EINVAL;
  if (a->world_size < 1 || a->n_panels < 1 || a->embedding_dim < 1 || a->embedding_dim % a->n_panels != 0) return HCSPMM_EINVAL;
  if (a->dtype < HCSPMM_DTYPE_F32 || a->dtype > HCSPMM_DTYPE_BF16) return HCSPMM_EINVAL;
  if (a->n_local < 0 || a->pad_rows < a->n_local || a->num_edges < 0 || !a->x_pm || !a->z_pm) return HCSPMM_EINVAL;
  if ((size_t)a->n_panels > c->landed.size()) return HCSPMM_EINVAL;
  const bool gather = a->world_size > 1 || a->always_gather != 0;
  if (gather && (!a->nccl_comm || !a->gathered_pm)) return HCSPMM_EINVAL;
  const int w = a->embedding_dim / a->n_panels;
  const size_t eb = elem_bytes(a->dtype);
  const int64_t x_rows = gather ? (int64_t)a->world_size * a->pad_rows : a->pad_rows;
  const size_t x_panel = (size_t)a->pad_rows * (size_t)w * eb, g_panel = (size_t)x_rows * (size_t)w * eb,
               z_panel = (size_t)a->n_local * (size_t)w * eb;
  hipStream_t compute = reinterpret_cast<hipStream_t>(a->compute_stream), comm = reinterpret_cast<hipStream_t>(a->comm_stream);
  const char* x = static_cast<const char*>(a->x_pm);
  char* g = static_cast<char*>(a->gathered_pm);
  char* z = static_cast<char*>(a->z_pm);
  if (gather) {
    // the gathers start behind everything the compute stream holds: the producers of x_pm and the previous step's
    // products, which still read gathered_pm
    hipError_t e = hipEventRecord(c->reached, compute);
    if (e == hipSuccess) e = hipStreamWaitEvent(comm, c->reached, 0);
    if (e != hipSuccess) return fail((int)e);
    for (int p = 0; p < a->n_panels; ++p) {
      const ncclResult_t r = ncclAllGather(x + (size_t)p * x_panel, g + (size_t)p * g_panel, (size_t)a->pad_rows * (size_t)w,
                                           nccl_type(a->dtype), reinterpret_cast<ncclComm_t>(a->nccl_comm), comm);
      if (r != ncclSuccess) return fail((int)r);
      e = hipEventRecord(c->landed[(size_t)p], comm);
      if (e != hipSuccess) return fail((int)e);
    }
  }
  for (int p = 0; p < a->n_panels; ++p) {
    if (gather) {
      const hipError_t e = hipStreamWaitEvent(compute, c->landed[(size_t)p], 0);  // gathers p + 1 .. keep running
      if (e != hipSuccess) return fail((int)e);
    }
    const void* xp = gather ? static_cast<const void*>(g + (size_t)p * g_panel) : static_cast<const void*>(x + (size_t)p * x_panel);
    const int rc = hcspmm_forward_typed(xp, x_rows, w, z + (size_t)p * z_panel, w, a->dtype, a->row_pointers_d, a->column_index_d,
                                        a->blockPartition_d, a->edgeToColumn_d, a->edgeToRow_d, a->hybrid_type_d, a->plan_d,
                                        a->plan_header_h, a->n_local, a->num_edges, w, a->workspace_d, a->workspace_bytes,
                                        a->compute_stream);
    if (rc != HCSPMM_OK) return rc;
  }
  return HCSPMM_OK;
}
