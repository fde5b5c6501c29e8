// update_kernels.hip -- the "update" half of the fused aggregate+update operators:
//   out[N x H] = agg[N x D] * W[D x H]      (fp32 in, fp32 MFMA, k-ascending fma chain)
// Replaces the second stage of the reference's fused kernels (hybrid_all_kernel.cu:1807-1837 in
// _32_fused, and the same block in _64_fused / _final_fused / _GIN_final_fused), which multiply
// the aggregated 16-row tile by the weight matrix with WMMA tf32.  Here: one wave per 16 rows,
// v_mfma_f32_16x16x4_f32, any D and H, arbitrary element strides for W (so the transposed view
// the reference's backward passes, GNN_model.py:98,120, is consumed without a copy).
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ------------------------------------------------------------------------------------------
// Streaming variant for the shapes GNN layers use (H = 16*T, T <= 4; D % 16 == 0; 16-byte aligned
// rows): the kernel is bound by reading agg once, so
//  * W is staged in LDS once per workgroup and the workgroup strides over many 64-row tiles;
//  * a lane loads agg with 16-byte accesses: lane (i = l & 15, kq = l >> 4) takes
//    agg[row i][16*kk + 4*kq .. +4), and the q-th of its four floats feeds k-step q -- the k order
//    inside a 16-chunk is a fixed permutation, applied to the W rows as well;
//  * tile t of a k-step multiplies by W columns T*j + t, so the T results of a lane are T
//    consecutive output columns and leave as one vector store.
// The accumulation order per output element is a fixed permutation of k (deterministic).
// ------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kUpdWaves * 64) void dense_update_stream_kernel(const float* __restrict__ in,
                                                                             const float* __restrict__ W,
                                                                             long long ldr, long long ldc,
                                                                             float* __restrict__ out, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [D][HS], HS = 16*T + 4 (row stride = 4 mod 8 words)
  constexpr int H = 16 * T;
  constexpr int HS = H + 4;
  for (int i = threadIdx.x; i < D * H; i += kUpdWaves * 64) {
    const int k = i / H, h = i - k * H;
    s_w[k * HS + h] = W[(long long)k * ldr + (long long)h * ldc];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int tiles = (N + 15) / 16;
  for (int tile = (int)blockIdx.x * kUpdWaves + wave; tile < tiles; tile += (int)gridDim.x * kUpdWaves) {
    const int r0 = tile * 16;
    const int row = r0 + i;
    const bool rok = row < N;
    const f32x4* arow = reinterpret_cast<const f32x4*>(in + (size_t)(rok ? row : 0) * (size_t)D) + kq;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += 64) {  // four 16-chunks of agg in flight
      f32x4 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rok && k0 + 16 * u < D) a[u] = arow[(k0 >> 2) + 4 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + 16 * u < D) {
          const float* wrow = s_w + (k0 + 16 * u + 4 * kq) * HS + T * i;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < T; ++t)
              acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q], wrow[q * HS + t], acc[t], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = r0 + 4 * kq + r;
      if (orow < N) {
        float* o = out + (size_t)orow * (size_t)H + T * i;
#pragma unroll
        for (int t = 0; t < T; ++t) o[t] = acc[t][r];
      }
    }
  }
}

template <int T>
static hipError_t launch_stream(const float* in, const float* W, long long ldr, long long ldc, float* out, int N, int D,
                                hipStream_t stream) {
  const size_t lds = (size_t)D * (16 * T + 4) * sizeof(float);
  const int tiles = (N + 15) / 16;
  int grid = (tiles + kUpdWaves - 1) / kUpdWaves;
  if (grid > 1024) grid = 1024;  // W is staged once per workgroup: stride over the row tiles
  hipLaunchKernelGGL((dense_update_stream_kernel<T>), dim3(grid), dim3(kUpdWaves * 64), lds, stream, in, W, ldr, ldc,
                     out, N, D);
  return hipGetLastError();
}

hipError_t launch_dense_update(const float* in, const float* W, long long ldr, long long ldc, float* out, int N,
                               int D, int H, hipStream_t stream) {
  if (N <= 0 || H <= 0) return hipSuccess;
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (aligned16 && D % 16 == 0 && H % 16 == 0 && H <= 64 && (size_t)D * (H + 4) * sizeof(float) <= 64 * 1024) {
    switch (H / 16) {
      case 1: return launch_stream<1>(in, W, ldr, ldc, out, N, D, stream);
      case 2: return launch_stream<2>(in, W, ldr, ldc, out, N, D, stream);
      case 3: return launch_stream<3>(in, W, ldr, ldc, out, N, D, stream);
      default: return launch_stream<4>(in, W, ldr, ldc, out, N, D, stream);
    }
  }
  const int rows_per_wg = 16 * kUpdWaves;
  const int grid = (N + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL(dense_update_kernel, dim3(grid), dim3(kUpdWaves * 64), 0, stream, in, W, ldr, ldc, out, N, D, H);
  return hipGetLastError();
}

}  // namespace hcspmm
