// fused_rows.hip -- fused aggregate + update with 16-row tiles of BOTH sub-paths (hcspmm_forward_fused, "row-tile form").
//
// The reference multiplies the aggregate of a window by the weights inside the aggregation launch for both window
// types (hybrid_all_kernel.cu:1639-1848 type-0 branch, :2067-2317, :2572-2770): one thread block owns a 16-row
// window, so its 16 aggregated rows are together in shared memory when the WMMA update starts.  Here the sparse-row
// path has no windows at launch time -- rows are tasks ordered by length class (plan_host.cpp) -- but the update
// GEMM does not care WHICH 16 rows share a tile: out[r] = z[r] * W is independent per row.  So a wave takes 16
// CONSECUTIVE TASKS of the length-sorted list (near-equal trip counts, exactly the property that makes the
// aggregation fast), sums them as the plain kernel does -- strictly in CSR order, the same bits -- writes each row of
// out2 from the registers, parks the 16 x D tile in a wave-private LDS area (one ds_write_b128 per lane and row),
// reads it back in MFMA A-operand order (ds_read_b128, conflict-free with a row stride of D + 4 words) and runs the
// same fp32 MFMA chain as dense_update_stream_kernel (same k permutation, W staged in LDS once per workgroup, same
// vector stores) => `out` has the two-launch form's bits without out2 ever being read back from HBM.
// No barrier after the weights are staged: a wave's LDS traffic is ordered with itself.
//
// Dense-tile windows produce the same kind of tile (the plain kernel's MFMA chain, accumulators written to out2 and to the LDS
// area) and take the same update.  Persistent workgroups stride over the tiles, so the weights are staged once per workgroup.
// Not handled here: rows summed by whole waves (wide tasks), split rows, column-sliced rows -- a few thousand rows, but up
// to a quarter of the entries: the hybrid kernel sums them as always (its ordinary / tiny / dense regions empty) and
// dense_update_rows_kernel multiplies those rows behind the fix-up pass.
// Applies when the sparse region is ONE column pass (D < 64, or short-row graphs at any D <= 128): a panel-major
// launch (Reddit-scale, D >= 64) never has a whole row in one wave, and there re-reading out2 costs 5 % of the step.
#include <mutex>

// k-steps per batch of the REGULAR dense chain (windows of more than 80 columns) in this unit only: the dense-tile kernel
// carries the update state next to the three chains, and the regular chain's batch sets its register allocation (spmm_impl.h's
// 8 is tuned for the plain kernel).  4 instead of 8: up to 9 % on the dense-heavy graphs at four waves per SIMD; five waves
// still spill 28-108 bytes and lose (profiles/r03/ab_fused_rows.log v11)
#ifndef HCSPMM_TILES_DENSE_B
#define HCSPMM_TILES_DENSE_B 4
#endif
#include "spmm_impl.h"

namespace hcspmm {

#ifndef HCSPMM_ROWS_MIN_WAVES
#define HCSPMM_ROWS_MIN_WAVES 5  // sparse-row tiles: 96 registers, nothing spilled; at 6 the gather batches spill 28-40 bytes per lane (-8 %), at 8 -28 %
#endif
#ifndef HCSPMM_DENSE_TILES_MIN_WAVES
#define HCSPMM_DENSE_TILES_MIN_WAVES 4
#endif

__host__ __device__ inline int rows_tile_stride(int D) { return D + 4; }  // words; 16 rows x 4 words hit 64 distinct banks
__host__ __device__ inline int rows_w_stride(int H) { return H + 4; }

// CSR-order sum of one task per lane group (the body of sparse_task for one column pass, without the store)
template <int L, int U>
__device__ __forceinline__ f32x4 rows_accumulate(const float* __restrict__ X, const int* __restrict__ col, int e0, int n,
                                                 size_t ldx, int csafe, bool cok, int lane) {
  const int s = lane & (L - 1);
  const int gbase = lane & ~(L - 1);
  int nmax = n;
#pragma unroll
  for (int off = L; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  f32x4 acc = azero<4>();
  int next = (s < n) ? col[e0 + s] : -1;
  for (int base = 0; base < nmax; base += L) {
    const int myidx = next;
    const bool more = base + L + s < n;
    next = -1;
    const int cnt = min(L, nmax - base);
    const int* pf = more ? col + e0 + base + L + s : nullptr;
    for (int j = 0; j < cnt;) {
      const int left = cnt - j;
      if (left > U / 2) {
        gather_batch<F32, 4, U>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
        j += U;
      } else if (U >= 8 && left > U / 4) {
        gather_batch<F32, 4, (U >= 8 ? U / 2 : 1)>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
        j += U / 2;
      } else if (U >= 4 && left > 1) {
        gather_batch<F32, 4, (U >= 8 ? U / 4 : 2)>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
        j += (U >= 8 ? U / 4 : 2);
      } else {
        gather_batch<F32, 4, 1>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
        j += 1;
      }
      pf = nullptr;
    }
  }
  return acc;
}

// a wave's own LDS writes must be visible to its own (cross-lane) reads: program order on the LDS pipe + no compiler motion
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the producers of a 16 x D tile (each also writes the rows of out2 = A*X it owns) ---------------------------------

// 16 consecutive ordinary tasks: 16/R rounds of R tasks, one lane group each
template <int L, int U>
__device__ __forceinline__ void ordinary_tile(const PlanArgs& a, const int4* __restrict__ tasks, int base, int ord_end,
                                              float* __restrict__ tile, int* __restrict__ trow, int TS, int lane, int col0, int width) {
  typedef Lane<F32, 4> Ln;
  constexpr int R = 64 / L;
  const float* X = reinterpret_cast<const float*>(a.X);
  float* Z = reinterpret_cast<float*>(a.Z);
  const int g = lane / L, s = lane & (L - 1);
  // column inside the chunk = column of the tile; a lane whose four columns would run past a width that is not a multiple of
  // 4 is moved back onto the last four (it then repeats its neighbour's values: spmm_impl.h lane_col)
  const bool cok = s * 4 < width;
  const int cl = cok ? lane_col<4>(s * 4, width) : 0;
  const int c = col0 + cl;   // feature column
  const int csafe = cok ? c : 0;
#pragma unroll 1
  for (int t = 0; t < 16 / R; ++t) {
    const int tid = base + t * R + g;
    int4 d = int4{-1, 0, 0, -1};
    if (tid < ord_end) d = tasks[tid];
    const f32x4 acc = rows_accumulate<L, U>(X, a.col, d.y, d.x >= 0 ? d.z : 0, a.ldx, csafe, cok, lane);
    if (cok && d.x >= 0) {
      if (d.w < 0) Ln::store(Z + (size_t)d.x * a.ldz + c, acc);
      else Ln::store_partial(a.partial + (size_t)d.w * (size_t)a.D + c, acc);
    }
    if (cok) *reinterpret_cast<typename MemF32<4>::type*>(tile + (t * R + g) * TS + cl) = acc;
    if (s == 0) trow[t * R + g] = (d.x >= 0 && d.w < 0) ? d.x : -1;  // a whole row: a row of out; segments wait for the fix-up pass
  }
}

// 16 consecutive tiny tasks (<= 2 entries, indices in the descriptor): T per lane group at a time, as tiny_tasks does
template <int L>
__device__ __forceinline__ void tiny_tile(const PlanArgs& a, const int4* __restrict__ tasks, int base, float* __restrict__ tile,
                                          int* __restrict__ trow, int TS, int lane, int col0, int width) {
  typedef Lane<F32, 4> Ln;
  constexpr int R = 64 / L;
  constexpr int T = R >= 16 ? 1 : 2;
  const float* X = reinterpret_cast<const float*>(a.X);
  float* Z = reinterpret_cast<float*>(a.Z);
  const int g = lane / L, s = lane & (L - 1);
  const bool cok = s * 4 < width;
  const int cl = cok ? lane_col<4>(s * 4, width) : 0;
  const int c = col0 + cl;
  const int csafe = cok ? c : 0;
#pragma unroll 1
  for (int st = 0; st < 16 / (R * T); ++st) {
    int4 d[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int tid = base + (st * T + t) * R + g;
      d[t] = (tid < a.n_tasks) ? tasks[tid] : int4{0, -1, -1, -1};  // .z < 0: no task
    }
    bool any1 = false, any2 = false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      any1 |= d[t].y >= 0;
      any2 |= d[t].w >= 0;
    }
    any1 = __builtin_amdgcn_ballot_w64(any1) != 0;
    any2 = __builtin_amdgcn_ballot_w64(any2) != 0;
    f32x4 v0[T], v1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) v0[t] = v1[t] = azero<4>();
    if (any1) {
#pragma unroll
      for (int t = 0; t < T; ++t) v0[t] = Ln::load(X + (size_t)max(d[t].y, 0) * a.ldx + csafe);
    }
    if (any2) {
#pragma unroll
      for (int t = 0; t < T; ++t) v1[t] = Ln::load(X + (size_t)max(d[t].w, 0) * a.ldx + csafe);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 acc = azero<4>();
      if (d[t].y >= 0) acc += v0[t];
      if (d[t].w >= 0) acc += v1[t];
      if (cok && d[t].z >= 0) {
        if (d[t].x >= 0) Ln::store(Z + (size_t)d[t].x * a.ldz + c, acc);
        else Ln::store_partial(a.partial + (size_t)(-(d[t].x + 1)) * (size_t)a.D + c, acc);
      }
      const int tr = (st * T + t) * R + g;
      if (cok) *reinterpret_cast<typename MemF32<4>::type*>(tile + tr * TS + cl) = acc;
      if (s == 0) trow[tr] = (d[t].z >= 0 && d[t].x >= 0) ? d[t].x : -1;
    }
  }
}

// one dense-tile window: the plain kernel's MFMA chain per panel of 16*DV columns (same bits), the C-layout accumulators
// written to Z and to the LDS tile (register r of lane (kq, j) is row 4*kq + r, columns c .. c + DV)
template <int DV>
__device__ __forceinline__ void dense_tile(const PlanArgs& a, int unit, float* __restrict__ tile, int* __restrict__ trow, int TS,
                                           int lane, int p0, int p1) {
  const float* X = reinterpret_cast<const float*>(a.X);
  float* Z = reinterpret_cast<float*>(a.Z);
  const int kq = lane >> 4, j = lane & 15;
  const int n_reg = a.n_dense - a.n_dense_compact - a.n_dense_compact2;
  int window;
#define HCSPMM_TILE_PANELS(CHAIN)                                                         \
  for (int panel = p0; panel < p1; ++panel) {                                              \
    const bool cok = panel * 16 * DV + j * DV < a.D;                                      \
    const int c = cok ? lane_col<DV>(panel * 16 * DV + j * DV, a.D) : 0;                  \
    const int csafe = c;                                                                  \
    f32x4 acc[DV];                                                                        \
    _Pragma("unroll") for (int q = 0; q < DV; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};    \
    CHAIN;                                                                                \
    if (cok) {                                                                            \
      dense_store<F32, DV>(Z, acc, window, kq, c, a.N, a.ldz);                            \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                     \
        typename AccT<DV>::type o;                                                        \
        _Pragma("unroll") for (int q = 0; q < DV; ++q) aset(o, q, acc[q][r]);             \
        *reinterpret_cast<typename MemF32<DV>::type*>(tile + (4 * kq + r) * TS + c - p0 * 16 * DV) = o; \
      }                                                                                   \
    }                                                                                     \
  }
  if (unit < n_reg) {
    cint_p dix = (cint_p)(a.plan + a.off_dense_index) + 4 * unit;  // wave-uniform: scalar loads
    window = dix[0];
    const int K4 = dix[2];
    const int* U = a.plan + a.off_dense_pack + dix[1];
    cu64_p masks = (cu64_p)(U + 4 * K4);
    HCSPMM_TILE_PANELS((dense_chain<F32, DV, false, HCSPMM_TILES_DENSE_B>(X, U, masks, K4, csafe, cok, a.ldx, lane, acc)))
  } else if (unit < n_reg + a.n_dense_compact2) {
    Rec<2> rec;
    const int* recp = a.plan + a.off_dense_compact2 + (unit - n_reg) * HCSPMM_COMPACT2_WORDS;
    rec.w[0] = recp[lane];
    rec.w[1] = recp[64 + lane];
    window = rec.template scalar<0>();
    const int K4 = rec.template scalar<1>();
    HCSPMM_TILE_PANELS((compact_chain<F32, DV, 2, false>(X, rec, K4, csafe, cok, a.ldx, lane, acc)))
  } else {
    Rec<1> rec;
    rec.w[0] = (a.plan + a.off_dense_compact + (unit - n_reg - a.n_dense_compact2) * HCSPMM_COMPACT_WORDS)[lane];
    window = rec.template scalar<0>();
    const int K4 = rec.template scalar<1>();
    HCSPMM_TILE_PANELS((compact_chain<F32, DV, 1, false>(X, rec, K4, csafe, cok, a.ldx, lane, acc)))
  }
#undef HCSPMM_TILE_PANELS
  if (lane < 16) trow[lane] = (window * 16 + lane < a.N) ? window * 16 + lane : -1;
}

// oacc += tile[:, 0 .. kcols) * W[kbase .. kbase + kcols, :]: the MFMA chain of dense_update_stream_kernel with the A operand read
// from LDS (s_wk = the staged weights from row kbase on); chunks in ascending k keep that kernel's order of additions
template <int HT>
__device__ __forceinline__ void tile_mac(const float* __restrict__ tile, const float* __restrict__ s_wk, int TS, int kcols, int lane,
                                         f32x4 (&oacc)[HT]) {
  constexpr int H = 16 * HT, HS = H + 4;
  const int i = lane & 15, kq = lane >> 4;
  const float* trd = tile + i * TS + 4 * kq;
#pragma unroll 2
  for (int k0 = 0; k0 < kcols; k0 += 16) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(trd + k0);
    const float* wrow = s_wk + (k0 + 4 * kq) * HS + HT * i;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int t = 0; t < HT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], wrow[q * HS + t], oacc[t], 0, 0, 0);
    }
  }
}

// Hreal = the operator's hidden width; 16*HT when it is a multiple of 16, else the staged weights carry zero columns up to
// 16*HT and the stores are masked (the update kernel's PAD form, update_kernels.hip: same chain, same bits in the real columns)
template <int HT>
__device__ __forceinline__ void tile_store(const int* __restrict__ trow, const f32x4 (&oacc)[HT], float* __restrict__ out, int lane,
                                           int Hreal) {
  constexpr int H = 16 * HT;
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = trow[4 * kq + r];
    if (orow >= 0) {
      float* o = out + (size_t)orow * (size_t)Hreal + HT * i;
      if (Hreal != H) {
        // pieces that do not fill 16-byte units: plain (cached) stores, so that the L2 assembles whole lines before they leave
        if constexpr (HT % 2 == 0) {
          if ((Hreal & 1) == 0) {
#pragma unroll
            for (int t = 0; t < HT; t += 2)
              if (HT * i + t < Hreal) *reinterpret_cast<typename MemF32<2>::type*>(o + t) = typename AccT<2>::type{oacc[t][r], oacc[t + 1][r]};
            continue;
          }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
          if (HT * i + t < Hreal) o[t] = oacc[t][r];
        continue;
      }
      // `out` is written once and not read by this operator: non-temporal stores (+4 ... +9 % on the whole call)
      if constexpr (HT == 4) {
        __builtin_nontemporal_store(f32x4{oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]}, reinterpret_cast<typename MemF32<4>::type*>(o));
      } else if constexpr (HT == 2) {
        __builtin_nontemporal_store(typename AccT<2>::type{oacc[0][r], oacc[1][r]}, reinterpret_cast<typename MemF32<2>::type*>(o));
      } else {
        __builtin_nontemporal_store(oacc[0][r], o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Persistent tile launches.  KIND 1: tiles of 16 ordinary tasks (heaviest first), then tiles of 16 tiny tasks; KIND 2: the
// dense-tile windows.  Each wave strides over the items of its launch (a sample of every length class per wave; the chip
// holds the whole grid at once), the weights are staged once per workgroup.  Two launches because their register needs
// differ: the sparse tiles fit 96 registers (five waves per SIMD, nothing spilled), the dense-window chains with the update
// state beside them spill 70-120 bytes at 96 and run at 128 (four waves); one launch for both -- and for the sliced and
// wide tasks as well -- was built and measured: 140 bytes of scratch at five waves (-25 %), +3.7 % / -4.4 % / +7.8 % on the
// TT / RD / YeastH-sized graphs at four, against +9.0 / +4.0 / +9.0 % for the split (profiles/r03/ab_fused_rows.log).
// ------------------------------------------------------------------------------------------
// WV waves per workgroup.  Embeddings wider than 64 columns (WV = 8, ta.chunk < D) are summed in column chunks of at most 64:
// the tile in LDS is 16 x chunk, the update accumulates `out` over the chunks in ascending k (the same order of additions),
// and eight waves share one staged copy of the weights -- a whole-width tile (16 x (D + 4) words per wave) next to W capped
// the sparse-tile launch at three workgroups per CU at D = 128.
template <int L, int HT, int DV, int KIND, int WV>
__global__ __launch_bounds__(64 * WV, (KIND == 1 && WV == 4) ? HCSPMM_ROWS_MIN_WAVES : HCSPMM_DENSE_TILES_MIN_WAVES) void fused_tiles_kernel(TilesArgs ta) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const PlanArgs& a = ta.p;
  constexpr int H = 16 * HT, HS = H + 4;
  constexpr int U = (L < HCSPMM_SPARSE_U) ? L : HCSPMM_SPARSE_U;
  const int TS = ta.tile_stride;
  float* s_w = s_mem;  // [D][HS]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // an input width that is not a multiple of 16 (the reference's 22 classes as the INPUT of the last layer's backward): the update
  // runs over Dp = the next multiple -- the staged weights get zero rows, the tile's columns [D, Dp) are zeroed once here and never
  // written again (the reference pads the same way, hybrid_all_kernel.cu:2748)
  const int Dp = (a.D + 15) & ~15;
  float* tile = s_mem + Dp * HS + wave * (16 * TS + 16);  // [16][TS] + 16 row ids
  int* trow = reinterpret_cast<int*>(tile + 16 * TS);
  for (int i = threadIdx.x; i < Dp * H; i += 64 * WV) {
    const int k = i / H, h = i - k * H;
    s_w[k * HS + h] = (h < a.H && k < a.D) ? a.W[(long long)k * a.w_ldr + (long long)h * a.w_ldc] : 0.0f;  // (zero columns / rows)
  }
  if (Dp != a.D) {
    const int pad = Dp - a.D;
    for (int i = lane; i < 16 * pad; i += 64) tile[(i / pad) * TS + a.D + (i - (i / pad) * pad)] = 0.0f;
  }
  __syncthreads();
  const int4* tasks = reinterpret_cast<const int4*>(a.plan + a.off_tasks);
  const int ord_end = a.n_tasks - a.n_tiny;
  const int n_items = KIND == 1 ? ta.n_ord_tiles + ta.n_tiny_tiles : a.n_dense;
  // column passes of one item of a chunked launch (WV = 8): sparse tiles ceil(D / chunk), dense windows one per panel
  [[maybe_unused]] const int n_pass = KIND == 1 ? (a.D + ta.chunk - 1) / ta.chunk : a.n_panels;
  // (Requesting the next item's compact record / tiny descriptors one item ahead -- two dependent round trips per tile instead
  // of three -- was built and measured: tiny tiles 3-4 % slower, dense windows -4 ... +5 %: profiles/r03/ab_fused_rows.log.)
  for (int item = (int)blockIdx.x * WV + wave; item < n_items; item += (int)gridDim.x * WV) {
    if constexpr (WV == kWaves) {
      // one pass: the tile holds whole rows, and the output accumulators live only while the update runs (kept alive across
      // the gathers they cost every one of these kernels 12-72 bytes of scratch)
      if constexpr (KIND == 1) {
        if (item < ta.n_ord_tiles) ordinary_tile<L, U>(a, tasks, a.n_wide + item * 16, ord_end, tile, trow, TS, lane, 0, a.D);
        else tiny_tile<L>(a, tasks, ord_end + (item - ta.n_ord_tiles) * 16, tile, trow, TS, lane, 0, a.D);
      } else {
        dense_tile<DV>(a, item, tile, trow, TS, lane, 0, a.n_panels);
      }
      wave_lds_fence();
      f32x4 oacc[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      tile_mac<HT>(tile, s_w, TS, Dp, lane, oacc);
      tile_store<HT>(trow, oacc, a.out, lane, a.H);
      wave_lds_fence();  // the next tile's writes stay behind these reads
    } else {
      f32x4 oacc[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int p = 0; p < n_pass; ++p) {
        int kbase, kcols;
        if constexpr (KIND == 1) {
          kbase = p * ta.chunk;
          kcols = min(ta.chunk, a.D - kbase);
          if (item < ta.n_ord_tiles) ordinary_tile<L, U>(a, tasks, a.n_wide + item * 16, ord_end, tile, trow, TS, lane, kbase, kcols);
          else tiny_tile<L>(a, tasks, ord_end + (item - ta.n_ord_tiles) * 16, tile, trow, TS, lane, kbase, kcols);
        } else {
          kbase = p * 16 * DV;
          kcols = min(16 * DV, a.D - kbase);
          dense_tile<DV>(a, item, tile, trow, TS, lane, p, p + 1);
        }
        wave_lds_fence();
        tile_mac<HT>(tile, s_w + kbase * HS, TS, kcols, lane, oacc);
        wave_lds_fence();  // the next chunk's writes stay behind these reads
      }
      tile_store<HT>(trow, oacc, a.out, lane, a.H);
    }
  }
}

// column chunk of the sparse tiles and waves per workgroup for an embedding width
static inline int tiles_chunk(int D) { return (D > 64 && D % 64 == 0) ? 64 : D; }  // (D = 96 in chunks of 48: 12 of 16 lanes busy, two walks of every task: -25 % on the TT-sized graph)
static inline int tiles_waves(int D) { return tiles_chunk(D) < D ? 8 : kWaves; }
static inline int tiles_stride(int D) { return (tiles_chunk(D) < D ? 64 : (D + 15) / 16 * 16) + 4; }  // (dense panels of a chunked launch are 64 wide)

static inline int tiles_ht(int H) { return (H + 15) / 16; }  // output tiles of 16 columns; widths in between are zero-padded
size_t fused_tiles_lds_bytes(int D, int H) {
  return ((size_t)((D + 15) / 16 * 16) * rows_w_stride(16 * tiles_ht(H)) + (size_t)tiles_waves(D) * (16 * tiles_stride(D) + 16)) * sizeof(float);
}

// shapes the row-tile form serves: fp32 rows of 17 to 128 columns (any width: the tile is zero-padded to the next multiple of 16), one lane group of at most 32 lanes per row, one, two or four
// output tiles (H <= 32 or 49 ... 64: the reference's default 22 classes pad to 32, as its own fused kernels pad to their
// tile, hybrid_all_kernel.cu:2748)
bool fused_tiles_supported(int D, int H) {
  return D >= 17 && D <= 128 && H >= 1 && H <= 64 && tiles_ht(H) != 3 && fused_tiles_lds_bytes(D, H) <= 64 * 1024;
}

// workgroups the chip holds at once (persistent launch: more than that would queue behind whole strided loops).  Asked once per
// kernel and LDS size (the occupancy query costs tens of microseconds on the host: not per launch).
template <int L, int HT, int DV, int KIND, int WV>
static long long resident_wgs(size_t lds) {
  static std::mutex mu;
  static size_t seen_lds[4] = {0, 0, 0, 0};
  static long long seen_wgs[4] = {0, 0, 0, 0};
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < 4; ++i)
    if (seen_wgs[i] > 0 && seen_lds[i] == lds) return seen_wgs[i];
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1024;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_tiles_kernel<L, HT, DV, KIND, WV>, 64 * WV, lds) != hipSuccess || per_cu <= 0)
    per_cu = 16 / WV;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const long long wgs = (long long)per_cu * cus;
  for (int i = 0; i < 4; ++i)
    if (seen_wgs[i] == 0) {
      seen_lds[i] = lds;
      seen_wgs[i] = wgs;
      break;
    }
  return wgs;
}

template <int L, int HT, int DV, int KIND, int WV>
static hipError_t launch_tiles_LHD(TilesArgs ta, hipStream_t stream) {
  constexpr int R = 64 / L;
  PlanArgs& a = ta.p;
  a.n_wide = (R > 1) ? a.n_wide : 0;
  a.dense_vec = DV;
  a.n_panels = (a.D + 16 * DV - 1) / (16 * DV);
  ta.chunk = tiles_chunk(a.D);
  ta.tile_stride = tiles_stride(a.D);
  ta.n_ord_tiles = (a.n_tasks - a.n_tiny - a.n_wide + 15) / 16;
  ta.n_tiny_tiles = (a.n_tiny + 15) / 16;
  const long long items = KIND == 1 ? (long long)ta.n_ord_tiles + ta.n_tiny_tiles : (long long)a.n_dense;
  if (items <= 0) return hipSuccess;
  static const long long forced = [] {
    const char* e = getenv("HCSPMM_ROWS_WGS");
    return e ? atoll(e) : 0LL;
  }();
  const size_t lds = fused_tiles_lds_bytes(a.D, a.H);
  // one resident round: every wave strides over the items, so each gets a sample of every class
  long long grid = (items + WV - 1) / WV;
  const long long cap = forced > 0 ? forced : resident_wgs<L, HT, DV, KIND, WV>(lds);
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((fused_tiles_kernel<L, HT, DV, KIND, WV>), dim3((unsigned)grid), dim3(64 * WV), lds, stream, ta);
  return hipGetLastError();
}

template <int L, int DV, int KIND, int WV>
static hipError_t launch_tiles_LDK(const TilesArgs& ta, hipStream_t stream) {
  switch (tiles_ht(ta.p.H)) {
    case 1: return launch_tiles_LHD<L, 1, DV, KIND, WV>(ta, stream);
    case 2: return launch_tiles_LHD<L, 2, DV, KIND, WV>(ta, stream);
    case 4: return launch_tiles_LHD<L, 4, DV, KIND, WV>(ta, stream);
    default: return hipErrorInvalidValue;
  }
}

template <int L, int DV, int WV>
static hipError_t launch_tiles_LD(const TilesArgs& ta, hipStream_t stream) {
  const hipError_t e = launch_tiles_LDK<L, DV, 1, WV>(ta, stream);
  if (e != hipSuccess || ta.p.n_dense <= 0) return e;
  return launch_tiles_LDK<L, DV, 2, WV>(ta, stream);
}

// The tile launches of the row-tile form: sparse-row tiles, then dense-window tiles.  a: as for the hybrid launch of the
// same call (X, Z = out2, partial, col, plan sections, n_wide / panel_cols from wide_choice, W / out / H); that launch
// (a.fused = 2: sliced and wide tasks only, then the fix-up pass) and dense_update_rows_kernel follow.
hipError_t launch_fused_tiles(const PlanArgs& a, hipStream_t stream) {
  if (!fused_tiles_supported(a.D, a.H) || a.panel_cols < a.D) return hipErrorInvalidValue;
  TilesArgs ta;
  ta.p = a;
  ta.n_ord_tiles = ta.n_tiny_tiles = ta.chunk = ta.tile_stride = 0;
  if (tiles_chunk(a.D) < a.D) return launch_tiles_LD<16, 4, 8>(ta, stream);  // D = 128: two chunks of 64
  switch (pick_L(a.D, 4)) {
    case 8: return launch_tiles_LD<8, 2, 4>(ta, stream);                                                       // D = 32
    case 16: return launch_tiles_LD<16, 4, 4>(ta, stream);  // D = 33 ... 64: one 64-column dense panel (two 32-column ones gather every row twice)
    case 32: return launch_tiles_LD<32, 4, 4>(ta, stream);                                                     // D = 80, 96, 112
    default: return hipErrorInvalidValue;
  }
}

}  // namespace hcspmm
