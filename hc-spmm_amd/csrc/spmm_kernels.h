// spmm_kernels.h -- internal launch interface between capi.hip and the kernel translation units
// (spmm_kernels.hip: fp32, spmm_kernels_h16.hip: fp16 / bf16; device code in spmm_impl.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hcspmm {

// Arguments of the planned hybrid launch (device pointers; `plan` is the uploaded blob of
// hcspmm_plan_build, the off_* / n_* fields are copied from the host copy of its header).
struct PlanArgs {
  const void* X;  // features, element type of the launch (fp32, or fp16 / bf16 bits)
  void* Z;
  float* partial;  // workspace: n_partials x D partial sums of split rows
  const int* col;
  const int* plan;
  size_t ldx, ldz;  // row strides of X and Z in elements (>= D)
  int off_tasks, n_tasks;
  int off_dense_index, off_dense_pack, n_dense;
  int off_dense_compact, n_dense_compact;    // the last n_dense_compact dense windows use fixed 64-word records
  int off_dense_compact2, n_dense_compact2;  // the n_dense_compact2 before them use 128-word records
  int off_fixups, n_split_rows;
  int N, D;
  int n_wide;                          // the n_wide longest tasks are summed by whole waves
  int n_tiny;                          // the last n_tiny tasks carry their (<= 2) indices inline
  int panel_cols;                      // feature columns per sparse pass (D = one pass; 32 = panel-major)
  int sparse_wgs_pp, dense_vec;        // filled by the launcher
  int wide_wgs, sparse_wgs, n_panels;  // filled by the launcher
  int tiny_wgs;                        // filled by the launcher: workgroups of the tiny-task region (per panel)
  int tiny_kernel_wgs;                 // filled by the launcher: > 0 = the tiny tasks run as their own launch (per panel)
  // XCD-affine column slices (hcspmm.h n_slices): slice s owns descriptors [table[s], table[s+1]) of the slice task
  // list and is served by the workgroups b = s (mod 8) of the sliced region, the first slice_wgs of every panel
  int off_slice_table, off_slice_tasks, n_slices, slice_xcd_tasks;
  int slice_wgs;                       // filled by the launcher (a multiple of 8)
  int free_wgs_pp;                     // filled by the launcher: wide + ordinary + tiny workgroups per panel
  // fused aggregate+update (hcspmm_forward_fused, fp32 only): when fused != 0 every dense-tile window also
  // multiplies its 16 x D tile by the weights while it is still in the MFMA accumulators and writes 16 rows of
  // out (N x H, row-major); Z is then the operator's out2.  W: D x H with element strides (w_ldr, w_ldc).
  int fused, H;
  int fused_dense_wgs;  // filled by the launcher: workgroups of the fused dense region (each strides over windows)
  const float* W;
  long long w_ldr, w_ldc;
  float* out;
};

// Arguments of the row-tile fused launch (fused_rows.hip): the planned launch's arguments (Z = out2; W / out / H set) plus the
// tile counts.  fp32, 16 bytes per lane.
struct TilesArgs {
  PlanArgs p;
  int n_ord_tiles, n_tiny_tiles;  // filled by the launcher: tiles of 16 ordinary / tiny tasks
  int chunk, tile_stride;         // filled by the launcher: feature columns per pass of a sparse tile (D: one pass), words per tile row
};

// Arguments of the plan-free launch: the reference's seven graph tensors as they are.
struct WindowArgs {
  const void* X;
  void* Z;
  const int* rowptr;
  const int* col;
  const int* blockPartition;
  const int* edgeToColumn;
  const int* edgeToRow;
  const int* hybrid_type;
  size_t ldx, ldz;  // row strides of X and Z in elements (>= D)
  int N, D;
};

// vec = elements per lane access.  fp32: 4 for every D >= 4 (element-aligned vectors: any stride, any base address), 2 / 1 for
// D = 2, 3 / 1.  16-bit: 8 (D >= 32) or 4 when D and the strides are even and the bases 4-byte aligned (dword-aligned vectors), else 1.
hipError_t launch_plan_f32(const PlanArgs& a, int vec, hipStream_t stream);
hipError_t launch_window_f32(const WindowArgs& a, int vec, hipStream_t stream);
hipError_t launch_plan_f16(const PlanArgs& a, int vec, hipStream_t stream);
hipError_t launch_window_f16(const WindowArgs& a, int vec, hipStream_t stream);
hipError_t launch_plan_bf16(const PlanArgs& a, int vec, hipStream_t stream);
hipError_t launch_window_bf16(const WindowArgs& a, int vec, hipStream_t stream);

// dW[D x H] = A^T * B (A: N x D rows lda apart, B: N x H rows ldb apart); `partial` holds
// weight_grad_groups(N) * D * H floats.  Shapes: weight_grad_supported(D, H).
bool weight_grad_supported(int D, int H);
int weight_grad_groups(long long N);
hipError_t launch_weight_grad(const float* A, long long lda, const float* B, long long ldb, float* out, float* partial,
                              long long N, int D, int H, hipStream_t stream);

// out[N x H] = in[N x D] * W (D x H, element strides ldr / ldc), fp32 MFMA.  tile_list (device, n_tiles ids of
// 16-row tiles) restricts the product to those tiles (the windows a fused launch has not already multiplied);
// nullptr = every tile.
hipError_t launch_dense_update(const float* in, const float* W, long long ldr, long long ldc, float* out, int N,
                               int D, int H, const int* tile_list, int n_tiles, hipStream_t stream);
// Row-tile form of the fused operators (fused_rows.hip): persistent launches that sum 16 tasks (or one dense window) at a
// time, write out2 and multiply the tile by the weights before it leaves the CU.  a: as for launch_plan_f32 with vec = 4
// (n_wide / panel_cols from wide_choice; needs panel_cols >= D); the hybrid launch with a.fused = 2 (sliced and wide tasks
// only + fix-up pass) follows.  fused_tiles_supported: the (D, H) shapes.
bool fused_tiles_supported(int D, int H);
hipError_t launch_fused_tiles(const PlanArgs& a, hipStream_t stream);
// out rows of the sparse-path rows that launch does not cover -- whole rows among the n_wide longest tasks, split rows
// (fix-up list), column-sliced rows that fell into one piece -- read from `in` (= out2, after the fix-up pass).
// Shapes: dense_update_streams(in, out, D, H).
hipError_t launch_dense_update_leftover(const float* in, const float* W, long long ldr, long long ldc, float* out, int N,
                                        int D, int H, const int* plan, int off_tasks, int n_wide, int off_fixups,
                                        int n_split_rows, int off_slice_tasks, int n_slice_tasks, hipStream_t stream);
// true when launch_dense_update takes the LDS-staged streaming kernel for this shape (the shapes the single-launch
// fused dense-tile epilogue is built for as well)
bool dense_update_streams(const float* in, const float* out, int D, int H);


}  // namespace hcspmm
