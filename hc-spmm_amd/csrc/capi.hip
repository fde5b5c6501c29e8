// capi.hip -- C-ABI entry points that enqueue device work (see include/hcspmm.h).
//
// hcspmm_forward replaces the reference launchers spmm_forward_plus / _more / _fixed32 / _fixed64
// (hybrid_kernel/hybrid_all_kernel.cu:410-594) and hcspmm_forward_fused the five fused launchers
// (:596-863).  Differences by design: work goes to the caller's stream (the reference uses the
// legacy default stream), every launch is checked (the reference checks none, :283-287), Z is the
// caller's buffer, and nothing synchronises or allocates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fingerprint.h"
#include "hcspmm.h"
#include "spmm_kernels.h"

namespace {
thread_local int g_last_hip_error = 0;

int fail_hip(hipError_t e) {
  g_last_hip_error = (int)e;
  return HCSPMM_EHIP;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

inline int elem_bytes(int dtype) { return dtype == HCSPMM_DTYPE_F32 ? 4 : 2; }

// Per-lane access width (in elements).  fp32: 16 bytes per lane for every embedding width of at least 4 columns, whatever
// the row strides and base addresses are -- the kernels address fp32 vectors with element alignment and move a lane whose
// columns would run past the row back onto the row's last four (spmm_impl.h MemF32 / lane_col) -- 8 bytes for D = 2, 3
// and single elements for D = 1.  16-bit features: below.
int pick_vec(int dtype, int D, int64_t ldx, int64_t ldz, const void* X, const void* Z, const void* ws) {
  // (8-byte lanes for fp32 widths up to 16 -- more lanes per row, eight loads in flight per lane -- were measured too: D = 16 is 10 %
  // SLOWER that way, D = 12 3 % faster, D = 4 ... 8 the same: not adopted)
  if (dtype == HCSPMM_DTYPE_F32) return D >= 4 ? 4 : D >= 2 ? 2 : 1;
  // 16-bit features: 8 (or, below 8 columns, 4) elements per lane whenever every row starts on a dword -- an even width, even row
  // strides, 4-byte aligned bases; the fp32 workspace is always that -- the last lane moved back onto the row's last elements;
  // single elements otherwise (odd widths or strides)
  const bool dword_rows = ((D | ldx | ldz) & 1) == 0 && aligned(X, 4) && aligned(Z, 4) && (!ws || aligned(ws, 4));
  // below 32 columns 8-byte lanes win: 5-8 lanes per row keep eight loads in flight per lane (L = 8), where three 16-byte lanes
  // (L = 4) keep four -- D = 20 / 22 / 24 are 7-10 % faster that way, D = 32 is not (profiles/r04/ab_ragged_lanes_bf16.log)
  if (dword_rows && D >= 32) return 8;
  if (dword_rows && D >= 4) return 4;
  return 1;
}

// the access width wide_choice assumes (from the embedding width alone, so that callers can ask ahead of a launch)
inline int nominal_vec(int dtype, int D) {
  if (dtype == HCSPMM_DTYPE_F32) return D >= 4 ? 4 : D >= 2 ? 2 : 1;
  return (D % 2 == 0 && D >= 32) ? 8 : (D % 2 == 0 && D >= 4) ? 4 : 1;
}
}  // namespace

extern "C" const char* hcspmm_strerror(int code) {
  switch (code) {
    case HCSPMM_OK: return "ok";
    case HCSPMM_EINVAL: return "invalid argument (null / negative size / column id out of range / X too short for the plan)";
    case HCSPMM_ENOMEM: return "host allocation failed";
    case HCSPMM_EPLAN: return "plan does not match this graph (magic/version/N/E/layout/fingerprint)";
    case HCSPMM_EHIP: return "HIP runtime error (see hcspmm_last_hip_error)";
    case HCSPMM_EWORKSPACE: return "workspace too small";
    case HCSPMM_ERANGE: return "size exceeds the int32 index contract";
    default: return "unknown hcspmm error";
  }
}

// Column-panel choice for the sparse region.  Wide embeddings are processed panel-major in slices of
// 32 fp32 columns (one 128-byte line per gathered row): with a quarter of every row in play at a
// time, four times as many distinct rows fit the per-XCD L2 -- measured at D = 32 the L2 hit rate is
// 51 % against 34 % at D = 128 on the same graph, and the launch is bound by what misses L2.  Each
// pass re-reads the column indices and pays the per-task overhead again, so short-row graphs
// (mean task length < 8) keep one pass over the full width.  HCSPMM_PANEL_COLS overrides
// (-1: one pass; n > 0: n columns, rounded up to a multiple of 16).
static int panel_choice(const hcspmm_plan_header* h, int D, int dtype) {
  static const int env = [] {
    const char* e = getenv("HCSPMM_PANEL_COLS");
    return e ? atoi(e) : 0;
  }();
  if (env < 0) return D;
  if (env > 0) return env >= D ? D : ((env + 15) / 16) * 16;
  if (h->panel_cols != 0) {  // the plan's own choice (hcspmm_plan_params.panel_cols; hcspmm.tune_plan measures it)
    const long long cols = h->panel_cols < 0 ? D : (long long)h->panel_cols * (4 / elem_bytes(dtype));
    return cols >= D ? D : (int)cols;
  }
  const int line_cols = 128 / elem_bytes(dtype);  // 32 fp32 or 64 16-bit columns: one cache line per gathered row
  if (D < 2 * line_cols || h->n_tasks <= 0) return D;
  const double mean_len = (double)h->nnz_sparse / ((double)h->n_tasks + (double)h->n_slice_tasks);
  if (mean_len < 8.0) return D;
  // an X beyond the 256 MiB Infinity Cache is gathered from HBM whatever the order: two lines per row and pass (256
  // contiguous bytes per gather) measure 2-3 % faster there than one (config 4 / 5 shares, 16 M-node power law:
  // profiles/r03/ab_panel_width.log), while a cache-resident X wants exactly one (Reddit-scale: 64 columns +8 %, all 128 +20 %)
  const double x_bytes = (double)h->num_columns * (double)D * (double)elem_bytes(dtype);
  const int cols = x_bytes > 256.0 * 1048576.0 ? 2 * line_cols : line_cols;
  return D >= 2 * cols ? cols : D;
}

// Wide-task threshold.  A lane group sums a task with U loads in flight, so a task of T entries is a
// chain of T/U dependent memory round trips; the whole launch is about
// passes * nnz_sparse / (8 * R * resident waves) such rounds deep.  The threshold is the largest
// power of two in [16, 256] not exceeding a quarter of that depth times 8: small (latency-bound)
// launches hand every row longer than 16 entries to a whole wave, large (throughput-bound) ones
// keep rows up to 256 entries on one lane group, in CSR order.
static int wide_choice(const hcspmm_plan_header* h, int D, int dtype, int* n_wide, int* panel_cols = nullptr) {
  const int vec = nominal_vec(dtype, D);
  const int pw = panel_choice(h, D, dtype);
  if (panel_cols) *panel_cols = pw;
  const double passes = (double)((D + pw - 1) / pw);
  int L = 4;
  while (L < (pw + vec - 1) / vec && L < 64) L <<= 1;
  const int R = 64 / L;
  *n_wide = 0;
  const double work = passes * (double)h->nnz_sparse;
  if (R == 1) return INT32_MAX;
  const double kResidentWaves = 256.0 * 16.0;
  const double t = 0.25 * work / ((double)R * kResidentWaves);
  int b = 0;
  while (b < 4 && (double)(16 << (b + 1)) <= t) ++b;
  *n_wide = h->n_len_gt[b];
  return *n_wide > 0 ? (16 << b) : INT32_MAX;
}

extern "C" int32_t hcspmm_wide_threshold_typed(const hcspmm_plan_header* h, int D, int dtype) {
  if (D <= 0 || dtype < HCSPMM_DTYPE_F32 || dtype > HCSPMM_DTYPE_BF16) return INT32_MAX;
  if (!h) {  // plan-free kernel: fixed threshold (kPlanFreeWide in spmm_impl.h) unless a wave holds one lane group
    const int vec = nominal_vec(dtype, D);
    return (D + vec - 1) / vec > 32 ? INT32_MAX : 64;
  }
  int n_wide = 0;
  return wide_choice(h, D, dtype, &n_wide);
}

extern "C" int32_t hcspmm_wide_threshold(const hcspmm_plan_header* h, int D) {
  return hcspmm_wide_threshold_typed(h, D, HCSPMM_DTYPE_F32);
}

extern "C" int hcspmm_abi_version(void) { return HCSPMM_ABI_VERSION; }

// Device variant of hcspmm_graph_fingerprint_host: every thread adds the terms of its grid-stride share, a wave
// folds them with shuffles and issues one 64-bit atomic add (the sum is order-independent, so the result is exact).
namespace {
__global__ __launch_bounds__(256) void fingerprint_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          long long N, long long E, unsigned long long* out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = (i0 == 0) ? hcspmm::fp_seed(N, E) : 0ull;
  for (long long r = i0; r <= N; r += stride) acc += hcspmm::fp_term_rowptr((uint64_t)r, rowptr[r]);
  for (long long e = i0; e < E; e += stride) acc += hcspmm::fp_term_col((uint64_t)e, col[e]);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
}  // namespace

namespace {
// edgeToRow[e] = largest r with rowptr[r] <= e (rows without entries never win: the search looks for the LAST such row)
__global__ __launch_bounds__(256) void edge_to_row_kernel(const int32_t* __restrict__ rowptr, int N, long long E,
                                                          int32_t* __restrict__ e2r) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    int lo = 0, hi = N;  // invariant: rowptr[lo] <= e < rowptr[hi]
    while (hi - lo > 1) {
      const int mid = lo + ((hi - lo) >> 1);
      if ((long long)rowptr[mid] <= e) lo = mid;
      else hi = mid;
    }
    e2r[e] = lo;
  }
}
}  // namespace

extern "C" int hcspmm_edge_to_row_device(const int32_t* rowptr, int64_t N, int64_t E, int32_t* e2r, void* stream_v) {
  if (N < 0 || E < 0 || !rowptr || (E > 0 && !e2r)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  if (E == 0) return HCSPMM_OK;
  long long blocks = (E + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(edge_to_row_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_v), rowptr,
                     (int)N, (long long)E, e2r);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}

extern "C" int hcspmm_graph_fingerprint_device(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E,
                                               uint64_t* out_d, void* stream_v) {
  if (!rowptr || !out_d || N < 0 || E < 0 || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  hipError_t e = hipMemsetAsync(out_d, 0, sizeof(uint64_t), stream);
  if (e != hipSuccess) return fail_hip(e);
  long long blocks = (N + 1 + E + 256 * 16 - 1) / (256 * 16);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rowptr, col, (long long)N, (long long)E,
                     reinterpret_cast<unsigned long long*>(out_d));
  e = hipGetLastError();
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}
extern "C" int hcspmm_last_hip_error(void) { return g_last_hip_error; }

namespace {
// weights / output of a fused launch (dense-tile windows multiply their tile by W inside the hybrid kernel)
struct FusedOperands {
  const float* W;
  long long ldr, ldc;
  float* out;
  int H;
  int mode;  // bit 0: dense-tile windows update inside the hybrid launch; bit 1: ordinary / tiny sparse rows in the row-tile launch
};

int forward_impl(const void* X, int64_t x_rows, int64_t ldx, void* Z, int64_t ldz, int dtype, const int32_t* rowptr,
                 const int32_t* col, const int32_t* blockPartition, const int32_t* edgeToColumn,
                 const int32_t* edgeToRow, const int32_t* hybrid_type, const int32_t* plan_d,
                 const hcspmm_plan_header* ph, int64_t N, int64_t E, int D, void* workspace,
                 size_t workspace_bytes, void* stream_v, const FusedOperands* fused) {
  if (dtype < HCSPMM_DTYPE_F32 || dtype > HCSPMM_DTYPE_BF16) return HCSPMM_EINVAL;
  if (N < 0 || E < 0 || D <= 0 || ldx < D || ldz < D) return HCSPMM_EINVAL;
  if (N == 0) return HCSPMM_OK;
  if (!X || !Z || !rowptr || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  hipError_t e;
  if (plan_d && ph) {
    const int rc = hcspmm_plan_check(ph, N, E, 0);
    if (rc != HCSPMM_OK) return rc;
    if (x_rows < ph->num_columns) return HCSPMM_EINVAL;  // the plan gathers rows X does not have
    const size_t need = hcspmm_workspace_bytes(ph, D);
    if (need > 0 && (!workspace || workspace_bytes < need)) return HCSPMM_EWORKSPACE;
    hcspmm::PlanArgs a;
    a.X = X;
    a.Z = Z;
    a.ldx = (size_t)ldx;
    a.ldz = (size_t)ldz;
    a.partial = need ? reinterpret_cast<float*>(workspace) : nullptr;
    a.col = col;
    a.plan = plan_d;
    a.off_tasks = ph->off_tasks;
    a.n_tasks = ph->n_tasks;
    a.n_tiny = ph->n_tiny;
    a.tiny_wgs = 0;
    a.tiny_kernel_wgs = 0;
    a.off_slice_table = ph->off_slice_table;
    a.off_slice_tasks = ph->off_slice_tasks;
    a.n_slices = ph->n_slices;
    a.slice_xcd_tasks = ph->slice_xcd_tasks;
    a.slice_wgs = 0;
    a.free_wgs_pp = 0;
    a.off_dense_index = ph->off_dense_index;
    a.off_dense_pack = ph->off_dense_pack;
    a.n_dense = ph->n_dense;
    a.off_dense_compact = ph->off_dense_compact;
    a.n_dense_compact = ph->n_dense_compact;
    a.off_dense_compact2 = ph->off_dense_compact2;
    a.n_dense_compact2 = ph->n_dense_compact2;
    a.off_fixups = ph->off_fixups;
    a.n_split_rows = ph->n_split_rows;
    wide_choice(ph, D, dtype, &a.n_wide, &a.panel_cols);
    a.sparse_wgs_pp = 0;
    a.dense_vec = 0;
    a.wide_wgs = 0;
    a.N = (int)N;
    a.D = D;
    a.sparse_wgs = 0;
    a.n_panels = 0;
    a.fused = fused ? fused->mode : 0;
    a.fused_dense_wgs = 0;
    a.H = fused ? fused->H : 0;
    a.W = fused ? fused->W : nullptr;
    a.w_ldr = fused ? fused->ldr : 0;
    a.w_ldc = fused ? fused->ldc : 0;
    a.out = fused ? fused->out : nullptr;
    const int vec = pick_vec(dtype, D, ldx, ldz, X, Z, need ? workspace : nullptr);
    if (fused && (vec != 4 || dtype != HCSPMM_DTYPE_F32)) return HCSPMM_EINVAL;  // (fused_form said otherwise)
    if (fused && (fused->mode & 2)) {
      // row-tile form: the ordinary / tiny tasks and the dense windows are summed AND multiplied by the tile launches
      // (fused_rows.hip); the hybrid launch below then runs the sliced and wide tasks only and ends with the fix-up pass,
      // which also adds the segments of split rows that sit among the ordinary tasks
      e = hcspmm::launch_fused_tiles(a, stream);
      if (e != hipSuccess) return fail_hip(e);
    }
    e = dtype == HCSPMM_DTYPE_F32 ? hcspmm::launch_plan_f32(a, vec, stream)
        : dtype == HCSPMM_DTYPE_F16 ? hcspmm::launch_plan_f16(a, vec, stream)
                                    : hcspmm::launch_plan_bf16(a, vec, stream);
  } else {
    if (fused) return HCSPMM_EINVAL;
    if (plan_d || ph) return HCSPMM_EINVAL;  // both or neither
    if (!blockPartition || !hybrid_type || (E > 0 && (!edgeToColumn || !edgeToRow))) return HCSPMM_EINVAL;
    hcspmm::WindowArgs a;
    a.X = X;
    a.Z = Z;
    a.ldx = (size_t)ldx;
    a.ldz = (size_t)ldz;
    a.rowptr = rowptr;
    a.col = col;
    a.blockPartition = blockPartition;
    a.edgeToColumn = edgeToColumn;
    a.edgeToRow = edgeToRow;
    a.hybrid_type = hybrid_type;
    a.N = (int)N;
    a.D = D;
    const int vec = pick_vec(dtype, D, ldx, ldz, X, Z, nullptr);
    e = dtype == HCSPMM_DTYPE_F32 ? hcspmm::launch_window_f32(a, vec, stream)
        : dtype == HCSPMM_DTYPE_F16 ? hcspmm::launch_window_f16(a, vec, stream)
                                    : hcspmm::launch_window_bf16(a, vec, stream);
  }
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}
}  // namespace

extern "C" int hcspmm_forward_typed(const void* X, int64_t x_rows, int64_t ldx, void* Z, int64_t ldz, int dtype, const int32_t* rowptr,
                                    const int32_t* col, const int32_t* blockPartition, const int32_t* edgeToColumn,
                                    const int32_t* edgeToRow, const int32_t* hybrid_type, const int32_t* plan_d,
                                    const hcspmm_plan_header* ph, int64_t N, int64_t E, int D, void* workspace,
                                    size_t workspace_bytes, void* stream_v) {
  return forward_impl(X, x_rows, ldx, Z, ldz, dtype, rowptr, col, blockPartition, edgeToColumn, edgeToRow, hybrid_type,
                      plan_d, ph, N, E, D, workspace, workspace_bytes, stream_v, nullptr);
}

extern "C" int hcspmm_forward_strided(const float* X, int64_t x_rows, int64_t ldx, float* Z, int64_t ldz, const int32_t* rowptr,
                                      const int32_t* col, const int32_t* blockPartition, const int32_t* edgeToColumn,
                                      const int32_t* edgeToRow, const int32_t* hybrid_type, const int32_t* plan_d,
                                      const hcspmm_plan_header* ph, int64_t N, int64_t E, int D, void* workspace,
                                      size_t workspace_bytes, void* stream_v) {
  return hcspmm_forward_typed(X, x_rows, ldx, Z, ldz, HCSPMM_DTYPE_F32, rowptr, col, blockPartition, edgeToColumn, edgeToRow,
                              hybrid_type, plan_d, ph, N, E, D, workspace, workspace_bytes, stream_v);
}

extern "C" int hcspmm_forward(const float* X, float* Z, const int32_t* rowptr, const int32_t* col,
                              const int32_t* blockPartition, const int32_t* edgeToColumn, const int32_t* edgeToRow,
                              const int32_t* hybrid_type, const int32_t* plan_d, const hcspmm_plan_header* ph,
                              int64_t N, int64_t E, int D, void* workspace, size_t workspace_bytes, void* stream_v) {
  return hcspmm_forward_strided(X, N, D, Z, D, rowptr, col, blockPartition, edgeToColumn, edgeToRow, hybrid_type, plan_d, ph,
                                N, E, D, workspace, workspace_bytes, stream_v);
}

// Forms of the fused operators (fp32).  0: two launches -- hybrid SpMM (out2 = A*X), then the streaming update over all rows.
// 1 ("in-launch", plans built with hcspmm_plan_params.fuse_in_launch = 1): the dense-tile windows multiply their tile by the
// weights while it is in the MFMA accumulators (spmm_impl.h fused_dense_region); the windows on the sparse-row path -- listed
// in the plan (off_sparse_windows) -- go through the update kernel afterwards.  Slower than 0 on MI355X (profiles/r02/ab_fused.log).
// 2 ("row-tile", fused_rows.hip): tiles of 16 consecutive tasks of the length-sorted list, and dense windows, are summed,
// written to out2, parked in LDS and multiplied before they leave the CU; the hybrid launch keeps the sliced and wide tasks,
// whose rows (a few thousand) a small launch multiplies behind the fix-up pass.  `out` has form 0's bits.  Needs the sparse
// region in ONE column pass (D < 64, or a short-row graph), 17 <= D <= 128, H <= 32 or 49 ... 64 (widths off the 16-column grid padded).
// Which one: HCSPMM_FUSED_SINGLE_LAUNCH=0 / 1 / 2 in the environment, else the plan's flags (fuse_in_launch = -1 / 1 / 2),
// else automatic: form 2 when out2 (N x D fp32) is 80 MB or more at D <= 64 -- it is then beyond what the update launch
// finds in the caches next to X and out, and not re-reading it is worth +2 ... +34 % (profiles/r03/ab_fused_rows.log: TT / RD /
// YeastH-sized low-degree graphs and dense-heavy graphs from 1/5 of their size up; every point above 80 MB gains, every
// sparse-row point below it loses 4-20 % -- the cache-resident Reddit-scale graph 1-2 %).  Beyond 64 columns (whole-row tiles
// up to 112 columns, two column chunks of 64 with eight-wave workgroups at 128) the tiles' LDS area costs occupancy: with
// H <= 32 graphs with a quarter of their rows in dense-tile windows gain 5-22 % (automatic from 256 MB of out2), sparse-row
// graphs -2 ... +4 %, and H = 64 is mixed (-6 ... +4 %), so those stay
// opt-in.  A shape outside a form falls back to the next one down.
static int fused_form(const hcspmm_plan_header* ph, const void* X, const void* out2, const void* out, int D, int H,
                      const void* workspace = nullptr) {
  static const int forced = [] {
    const char* e = getenv("HCSPMM_FUSED_SINGLE_LAUNCH");
    return !e ? -1 : (e[0] == '0' ? 0 : (e[0] == '2' ? 2 : 1));
  }();
  if (!ph) return 0;
  int asked = forced;
  if (asked < 0) {
    if (ph->flags & HCSPMM_PLAN_FUSE_NEVER) asked = 0;
    else if (ph->flags & HCSPMM_PLAN_FUSE_ROWS) asked = 2;
    else if (ph->flags & HCSPMM_PLAN_FUSE_IN_LAUNCH) asked = 1;
    else {
      const double out2_bytes = (double)ph->num_nodes * (double)D * 4.0;
      const bool dense_heavy = 64.0 * (double)ph->n_dense >= (double)ph->num_nodes;  // a quarter of the rows in dense-tile windows
      asked = (D <= 64 ? out2_bytes >= 80e6 : (D <= 128 && H <= 32 && dense_heavy && out2_bytes >= 256e6)) ? 2 : 0;
      // (both widths off the 16-column grid -- 22 x 22 -- is the one padded shape that measured slower than two launches on the
      // sparse-row graphs, -9 ... -12 %: profiles/r04/ab_dense_panels.log; it stays opt-in)
      if (D % 16 != 0 && H % 16 != 0) asked = 0;
    }
  }
  if (asked == 0) return 0;
  // both forms are 16-byte-per-lane builds: a caller's workspace that is only 4- or 8-byte aligned takes the two-launch form
  // (which serves every alignment) instead of failing
  if (!aligned(X, 16) || !aligned(out2, 16) || (workspace && !aligned(workspace, 16))) return 0;
  if (asked >= 2 && hcspmm::fused_tiles_supported(D, H) && aligned(out, 16) && panel_choice(ph, D, HCSPMM_DTYPE_F32) >= D) return 2;
  if (!(ph->flags & HCSPMM_PLAN_FUSE_IN_LAUNCH) && forced != 1) return 0;  // form 1 only where it was asked for by name
  bool dense_ok = ph->n_dense > 0 && D % 16 == 0 && D >= 32 && H % 16 == 0 && H <= 32 && H > 0;
  if (dense_ok) {
    const int dv = D <= 32 ? 2 : 4;  // (launch_plan_LV: the narrowest lane width that covers the row in one panel)
    const int rows = (D + 16 * dv - 1) / (16 * dv) * 16 * dv;
    dense_ok = (size_t)rows * (size_t)(H + 4) * sizeof(float) <= 64 * 1024;
  }
  return dense_ok ? 1 : 0;
}

extern "C" int hcspmm_fused_in_launch(const hcspmm_plan_header* ph, int D, int H) {
  return fused_form(ph, nullptr, nullptr, nullptr, D, H);  // (null pointers count as aligned)
}

extern "C" int hcspmm_forward_fused(const float* X, float* out, float* out2, const float* weights, int64_t ldr,
                                    int64_t ldc, int H, const int32_t* rowptr, const int32_t* col,
                                    const int32_t* blockPartition, const int32_t* edgeToColumn,
                                    const int32_t* edgeToRow, const int32_t* hybrid_type, const int32_t* plan_d,
                                    const hcspmm_plan_header* ph, int64_t N, int64_t E, int D, void* workspace,
                                    size_t workspace_bytes, void* stream_v) {
  if (!out || !out2 || !weights || H <= 0) return HCSPMM_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
  const int form = (plan_d && ph) ? fused_form(ph, X, out2, out, D, H, hcspmm_workspace_bytes(ph, D) ? workspace : nullptr) : 0;
  if (form == 2) {
    const FusedOperands f{weights, (long long)ldr, (long long)ldc, out, H, 2};
    const int rc = forward_impl(X, N, D, out2, D, HCSPMM_DTYPE_F32, rowptr, col, blockPartition, edgeToColumn, edgeToRow,
                                hybrid_type, plan_d, ph, N, E, D, workspace, workspace_bytes, stream_v, &f);
    if (rc != HCSPMM_OK) return rc;
    int n_wide = 0;
    wide_choice(ph, D, HCSPMM_DTYPE_F32, &n_wide);
    const hipError_t e = hcspmm::launch_dense_update_leftover(out2, weights, (long long)ldr, (long long)ldc, out, (int)N, D, H,
                                                              plan_d, ph->off_tasks, n_wide, ph->off_fixups, ph->n_split_rows,
                                                              ph->off_slice_tasks, ph->n_slice_tasks, stream);
    return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
  }
  if (form == 1) {
    const FusedOperands f{weights, (long long)ldr, (long long)ldc, out, H, 1};
    const int rc = forward_impl(X, N, D, out2, D, HCSPMM_DTYPE_F32, rowptr, col, blockPartition, edgeToColumn, edgeToRow,
                                hybrid_type, plan_d, ph, N, E, D, workspace, workspace_bytes, stream_v, &f);
    if (rc != HCSPMM_OK) return rc;
    if (ph->n_sparse_windows == 0) return HCSPMM_OK;
    const hipError_t e = hcspmm::launch_dense_update(out2, weights, (long long)ldr, (long long)ldc, out, (int)N, D, H,
                                                     plan_d + ph->off_sparse_windows, ph->n_sparse_windows, stream);
    return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
  }
  const int rc = hcspmm_forward(X, out2, rowptr, col, blockPartition, edgeToColumn, edgeToRow, hybrid_type, plan_d, ph,
                                N, E, D, workspace, workspace_bytes, stream_v);
  if (rc != HCSPMM_OK) return rc;
  const hipError_t e = hcspmm::launch_dense_update(out2, weights, (long long)ldr, (long long)ldc, out, (int)N, D, H, nullptr, 0,
                                                   stream);
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}

extern "C" int hcspmm_dense_update(const float* in, const float* weights, int64_t ldr, int64_t ldc, float* out, int64_t N, int D,
                                   int H, void* stream_v) {
  if (N < 0 || D <= 0 || H <= 0 || (N > 0 && (!in || !weights || !out))) return HCSPMM_EINVAL;
  if (N > INT32_MAX) return HCSPMM_ERANGE;
  const hipError_t e = hcspmm::launch_dense_update(in, weights, (long long)ldr, (long long)ldc, out, (int)N, D, H, nullptr, 0,
                                                   reinterpret_cast<hipStream_t>(stream_v));
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}

extern "C" size_t hcspmm_weight_grad_workspace(int64_t N, int D, int H) {
  if (N <= 0 || !hcspmm::weight_grad_supported(D, H)) return 0;
  return (size_t)hcspmm::weight_grad_groups(N) * (size_t)D * (size_t)H * sizeof(float);
}

extern "C" int hcspmm_weight_grad(const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t N, int D,
                                  int H, void* workspace, size_t workspace_bytes, void* stream_v) {
  if (N <= 0 || D <= 0 || H <= 0 || lda < D || ldb < H || !A || !B || !out) return HCSPMM_EINVAL;
  if (!hcspmm::weight_grad_supported(D, H)) return HCSPMM_EINVAL;
  if (N > INT32_MAX) return HCSPMM_ERANGE;
  if (!workspace || workspace_bytes < hcspmm_weight_grad_workspace(N, D, H)) return HCSPMM_EWORKSPACE;
  const hipError_t e = hcspmm::launch_weight_grad(A, lda, B, ldb, out, reinterpret_cast<float*>(workspace), N, D, H,
                                                  reinterpret_cast<hipStream_t>(stream_v));
  return e == hipSuccess ? HCSPMM_OK : fail_hip(e);
}
