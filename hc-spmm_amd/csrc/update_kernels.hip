// update_kernels.hip -- the "update" half of the fused aggregate+update operators:
//   out[N x H] = agg[N x D] * W[D x H]      (fp32 in, fp32 MFMA, k-ascending fma chain)
// Replaces the second stage of the reference's fused kernels (hybrid_all_kernel.cu:1807-1837 in
// _32_fused, and the same block in _64_fused / _final_fused / _GIN_final_fused), which multiply
// the aggregated 16-row tile by the weight matrix with WMMA tf32.  Here: one wave per 16 rows,
// v_mfma_f32_16x16x4_f32, any D and H, arbitrary element strides for W (so the transposed view
// the reference's backward passes, GNN_model.py:98,120, is consumed without a copy).
#include <hip/hip_runtime.h>

#include "spmm_kernels.h"

namespace hcspmm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kUpdWaves = 4;
constexpr int kMaxTiles = 8;  // 16-column output tiles kept in registers per pass (128 columns)

__global__ __launch_bounds__(kUpdWaves * 64) void dense_update_kernel(const float* __restrict__ in,
                                                                      const float* __restrict__ W, long long ldr,
                                                                      long long ldc, float* __restrict__ out, int N,
                                                                      int D, int H) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r0 = ((int)blockIdx.x * kUpdWaves + wave) * 16;
  if (r0 >= N) return;
  const int i = lane & 15, kq = lane >> 4;
  const int row = r0 + i;
  const bool rok = row < N;
  const float* arow = in + (size_t)(rok ? row : 0) * (size_t)D;
  for (int h0 = 0; h0 < H; h0 += 16 * kMaxTiles) {
    const int ntiles = min(kMaxTiles, (H - h0 + 15) / 16);
    f32x4 acc[kMaxTiles];
#pragma unroll
    for (int t = 0; t < kMaxTiles; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += 4) {
      const int k = k0 + kq;
      const bool kok = k < D;
      const float av = (rok && kok) ? arow[k] : 0.0f;
#pragma unroll
      for (int t = 0; t < kMaxTiles; ++t) {
        if (t < ntiles) {
          const int n = h0 + t * 16 + i;  // B operand: lane (k = kq, n = lane & 15)
          const float bv = (kok && n < H) ? W[(long long)k * ldr + (long long)n * ldc] : 0.0f;
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < kMaxTiles; ++t) {
      if (t < ntiles) {
        const int n = h0 + t * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = r0 + 4 * kq + r;
          if (orow < N && n < H) out[(size_t)orow * (size_t)H + n] = acc[t][r];
        }
      }
    }
  }
}

hipError_t launch_dense_update(const float* in, const float* W, long long ldr, long long ldc, float* out, int N,
                               int D, int H, hipStream_t stream) {
  if (N <= 0 || H <= 0) return hipSuccess;
  const int rows_per_wg = 16 * kUpdWaves;
  const int grid = (N + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL(dense_update_kernel, dim3(grid), dim3(kUpdWaves * 64), 0, stream, in, W, ldr, ldc, out, N, D, H);
  return hipGetLastError();
}

}  // namespace hcspmm
