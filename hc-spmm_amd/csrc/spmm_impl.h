// spmm_impl.h -- gfx950 (MI355X, CDNA4) device code of the hybrid SpMM  Z = A * X, templated on the
// feature element type (fp32; fp16 / bf16 features with fp32 accumulation -- paper p.16 Table VII).
// Included by spmm_kernels.hip (fp32 instantiations) and spmm_kernels_h16.hip (16-bit ones).
//
// Replaces the reference kernels spmm_forward_cuda_kernel_arbi_warps_hybrid_{adaptive,32,64,
// adaptive_more} (hybrid_kernel/hybrid_all_kernel.cu:919-1637).  Nothing here is derived from
// their structure: the reference runs one 96-thread block per 16-row window with warp-per-row
// gathers and WMMA tf32 tiles staged through shared memory; this file is wave64 code built
// around two facts of the machine (see /DESIGN.md):
//
//  * sparse-row path: the launch is bound by the gathered X-row bytes that miss the per-XCD L2, so
//    (a) rows become *tasks* ordered by power-of-two length class (host plan); L lanes of a wave own
//    one task and walk its neighbours strictly in CSR order (bit-identical to a sequential fp32
//    sum), 64/L tasks share a wave; the longest tasks go to whole waves ("wide", shuffle-tree
//    combine); (b) wide embeddings are processed panel-major, one cache line (128 bytes) per row
//    at a time across the whole grid; (c) column indices are fetched coalesced once per L
//    neighbours and broadcast through the LDS crossbar (ds_bpermute); every lane issues 16-byte
//    loads in branch-free batches of 8; (d) rows of at most two entries -- most rows of a low-degree
//    graph -- are "tiny" tasks whose column indices travel inside the task descriptor (their own launch, at eight
//    waves per SIMD, when there are enough of them); (e) rows longer than 256 entries are cut into XCD-affine column
//    slices: slice s is served only by workgroups b = s (mod 8), which share one XCD's L2, so that L2 sees 1/8 of the
//    columns from those tasks -- the headline went from fabric-bound (52 % L2 hits) to L2-gather-bound (67 %).
//  * dense-tile path: v_mfma_f32_16x16x4_f32 takes its B operand one fp32 per lane, so the
//    gathered X rows go from HBM straight into MFMA operand registers with 16-byte loads (the
//    VEC elements of a lane feed VEC MFMAs whose results re-assemble into one vector store);
//    the 0/1 tile of A arrives as a 64-bit lane mask per k-step, packed by the host in MFMA
//    lane order.  No LDS round trip, no barrier.  fp32 MFMA is an exact k-ordered fma chain,
//    so the result equals the sequential sum over the window's ascending unique columns.
//    Windows of at most 80 columns are described by fixed-size records addressed by unit number
//    (one or two coalesced loads, then the gathers); wider ones by an index entry + a pack.
//
// 16-bit features: rows are gathered as they are stored (half the bytes), widened to fp32 in
// registers (exact), summed in fp32 in the same order as the fp32 path, and rounded once (RNE) when
// the row is stored.  The dense-tile path keeps the fp32 MFMA (the 16-bit MFMAs want four k-values
// of one column per lane -- a transposed gather).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "hcspmm.h"
#include "spmm_kernels.h"

namespace hcspmm {

// ---------------------------------------------------------------- element types and lane vectors
struct F32 { typedef float T; };
struct F16 { typedef unsigned short T; };   // IEEE binary16 bits
struct BF16 { typedef unsigned short T; };  // bfloat16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// fp32 accumulator of VEC feature columns
template <int VEC> struct AccT { typedef float type __attribute__((ext_vector_type(VEC))); };
template <> struct AccT<1> { typedef float type; };
__device__ __forceinline__ float aget(const float& v, int) { return v; }
__device__ __forceinline__ void aset(float& v, int, float x) { v = x; }
template <typename V> __device__ __forceinline__ float aget(const V& v, int i) { return v[i]; }
template <typename V> __device__ __forceinline__ void aset(V& v, int i, float x) { v[i] = x; }
template <int VEC> __device__ __forceinline__ typename AccT<VEC>::type azero() {
  typename AccT<VEC>::type z;
#pragma unroll
  for (int q = 0; q < VEC; ++q) aset(z, q, 0.f);
  return z;
}

// VEC elements as they sit in memory
template <typename E, int VEC> struct RawT { typedef typename AccT<VEC>::type type; };  // F32: the floats themselves
template <> struct RawT<F16, 8> { typedef u32x4 type; };
template <> struct RawT<F16, 4> { typedef u32x2 type; };
template <> struct RawT<F16, 2> { typedef unsigned int type; };
template <> struct RawT<F16, 1> { typedef unsigned short type; };
template <> struct RawT<BF16, 8> { typedef u32x4 type; };
template <> struct RawT<BF16, 4> { typedef u32x2 type; };
template <> struct RawT<BF16, 2> { typedef unsigned int type; };
template <> struct RawT<BF16, 1> { typedef unsigned short type; };

template <typename E> __device__ __forceinline__ float widen(unsigned short h);
template <> __device__ __forceinline__ float widen<F16>(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
template <> __device__ __forceinline__ float widen<BF16>(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
template <typename E> __device__ __forceinline__ unsigned short narrow(float f);
template <> __device__ __forceinline__ unsigned short narrow<F16>(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
template <> __device__ __forceinline__ unsigned short narrow<BF16>(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__device__ __forceinline__ unsigned rword(const u32x4& v, int i) { return v[i]; }
__device__ __forceinline__ unsigned rword(const u32x2& v, int i) { return v[i]; }
__device__ __forceinline__ unsigned rword(const unsigned& v, int) { return v; }
__device__ __forceinline__ void rset(u32x4& v, int i, unsigned w) { v[i] = w; }
__device__ __forceinline__ void rset(u32x2& v, int i, unsigned w) { v[i] = w; }
__device__ __forceinline__ void rset(unsigned& v, int, unsigned w) { v = w; }

// fp32 vectors as they are addressed in memory: ELEMENT-aligned.  gfx950's global_load / global_store_dwordx4 need dword
// alignment only, and saying so lets one 16-byte-per-lane build serve every embedding width of at least 4 columns: a lane
// whose 4 columns would run past the row is moved back to the row's last 4 (lane_col), so it overlaps its neighbour --
// both then hold the same sums of the same rows in the same order and store the same bits twice.  The reference packs
// the remainder columns of such widths into spare warp lanes instead (hybrid_all_kernel.cu:996-1036; paper Table III).
template <int VEC> struct MemF32 { typedef float type __attribute__((ext_vector_type(VEC), aligned(4))); };
template <> struct MemF32<1> { typedef float type; };

// 16-bit rows as they are addressed in memory: DWORD-aligned vectors of packed pairs.  An even width with even row strides keeps
// every row on the dword grid, so the same trick serves fp16 / bf16 widths that are not multiples of 8 (the last lane moved back
// onto the row's last 8 elements): D = 22 takes three 16-byte lanes instead of 22 two-byte ones.
template <int WORDS> struct MemU32 { typedef unsigned int type __attribute__((ext_vector_type(WORDS), aligned(4))); };
template <> struct MemU32<1> { typedef unsigned int type; };

// first of the VEC feature columns a lane covers, given the aligned position c < cend of its slot in [.., cend)
template <int VEC> __device__ __forceinline__ int lane_col(int c, int cend) { return VEC == 1 ? c : min(c, cend - VEC); }

template <typename E, int VEC> struct Lane {
  typedef typename E::T T;
  typedef typename RawT<E, VEC>::type raw_t;
  typedef typename AccT<VEC>::type acc_t;
  static __device__ __forceinline__ raw_t zero() {
    raw_t z;
    __builtin_memset(&z, 0, sizeof(z));
    return z;
  }
  static __device__ __forceinline__ raw_t load(const T* p) {
    if constexpr (sizeof(T) == 4) return *reinterpret_cast<const typename MemF32<VEC>::type*>(p);
    else if constexpr (VEC >= 2) return *reinterpret_cast<const typename MemU32<VEC / 2>::type*>(p);
    else return *reinterpret_cast<const raw_t*>(p);
  }
  static __device__ __forceinline__ acc_t load_partial(const float* p) { return *reinterpret_cast<const typename MemF32<VEC>::type*>(p); }
  // element q of a loaded vector, widened (exact)
  static __device__ __forceinline__ float elem(const raw_t& v, int q) {
    if constexpr (sizeof(T) == 4) return aget(v, q);
    else if constexpr (VEC == 1) return widen<E>(v);
    else {
      const unsigned w = rword(v, q >> 1);
      return widen<E>((unsigned short)((q & 1) ? (w >> 16) : (w & 0xffffu)));
    }
  }
  static __device__ __forceinline__ void add(acc_t& acc, const raw_t& v) {
    if constexpr (sizeof(T) == 4) acc += v;
    else {
#pragma unroll
      for (int q = 0; q < VEC; ++q) aset(acc, q, aget(acc, q) + elem(v, q));
    }
  }
  static __device__ __forceinline__ raw_t pack(const acc_t& acc) {  // one rounding per element (16-bit types)
    if constexpr (sizeof(T) == 4) return acc;
    else if constexpr (VEC == 1) return narrow<E>(acc);
    else {
      raw_t r;
#pragma unroll
      for (int w = 0; w < VEC / 2; ++w)
        rset(r, w, (unsigned)narrow<E>(aget(acc, 2 * w)) | ((unsigned)narrow<E>(aget(acc, 2 * w + 1)) << 16));
      return r;
    }
  }
  // Z rows are written once and not re-read by this launch: non-temporal stores (0-5 %, profiles/r01/ab_nt_store.log)
  static __device__ __forceinline__ void store(T* p, const acc_t& acc) {
    if constexpr (sizeof(T) == 4) __builtin_nontemporal_store(pack(acc), reinterpret_cast<typename MemF32<VEC>::type*>(p));
    else if constexpr (VEC >= 2) __builtin_nontemporal_store(pack(acc), reinterpret_cast<typename MemU32<VEC / 2>::type*>(p));
    else __builtin_nontemporal_store(pack(acc), reinterpret_cast<raw_t*>(p));
  }
  static __device__ __forceinline__ void store_partial(float* p, const acc_t& acc) {
    __builtin_nontemporal_store(acc, reinterpret_cast<typename MemF32<VEC>::type*>(p));
  }
};

// The plan is read-only for the whole launch.  Wave-uniform reads of it (dense index entries, tile masks) go
// through the constant address space so that they are scalar loads (s_load_*) whatever the surrounding
// control flow: when that was left to the compiler's no-clobber analysis, adding the compact-record branch
// silently turned them into vector loads and cost the regular dense units 16 % (profiles/r01/ab_scalar_masks.log).
#define HCSPMM_CONST_AS __attribute__((address_space(4)))
typedef const HCSPMM_CONST_AS int* cint_p;
typedef const HCSPMM_CONST_AS unsigned long long* cu64_p;

constexpr int kWaves = 4;            // waves per workgroup (256 threads)
constexpr int kThreads = kWaves * 64;
#ifndef HCSPMM_SPARSE_U
#define HCSPMM_SPARSE_U 8  // row loads in flight per lane on the sparse-row path
#endif
#ifndef HCSPMM_DENSE_B
#define HCSPMM_DENSE_B 8   // k-steps (16-byte row loads in flight per lane) per batch on the dense-tile path
#endif
#ifndef HCSPMM_TINY_PER_WAVE
#define HCSPMM_TINY_PER_WAVE 8  // tiny tasks (<= 2 entries) per wave: T = 8 / (64/L) per lane group, at least 2, at most 4
#endif                          // (D = 32: T = 2 vs 4 vs 8 in profiles/r01/ab_tiny_tasks.log; wide D: ab_tiny_tasks_wide.log;
                                //  T = 8 costs the L = 32 build its fifth wave per SIMD: 119 instead of 94 registers)
template <int L> struct TinyT {
  static constexpr int per_group = HCSPMM_TINY_PER_WAVE / (64 / L);
  static constexpr int value = per_group < 2 ? 2 : (per_group > 4 ? 4 : per_group);
};
#ifndef HCSPMM_MIN_WAVES_H16
#define HCSPMM_MIN_WAVES_H16 4  // fp16 / bf16 builds (8 elements per lane, widened in registers): 119 registers; at 96 they spill 150+ bytes
#endif
#ifndef HCSPMM_FUSED_MIN_WAVES
#define HCSPMM_FUSED_MIN_WAVES 4
#endif
#ifndef HCSPMM_MIN_WAVES_PER_SIMD
#define HCSPMM_MIN_WAVES_PER_SIMD 5  // <= 96 registers per lane: five waves per SIMD.  Round 1 asked for four and got five by luck (94 registers);
                                     // when round 2's refactoring nudged the kernel to 100 registers the low-degree workloads lost 10 %
                                     // (profiles/r02/ab_occupancy.log), so the bound is now explicit
#endif

// One branch-free batch of UB row gathers: every lane issues all UB loads (finished tasks and lanes
// beyond the embedding width re-read row 0 / column 0, an L1 hit, and discard it), so the batch is one
// basic block -- UB broadcasts, UB address computations, UB loads back to back, counted waits.
// (Staging the rows through LDS instead was built and measured -- never faster, +9 % at D = 32:
// profiles/r01/ab_lds_stage.log, code at commit 950912e.)
template <typename E, int VEC, int UB>
__device__ __forceinline__ void gather_batch(const typename E::T* __restrict__ X, size_t ldx, int csafe, bool cok,
                                             int myidx, int src0, typename AccT<VEC>::type& acc,
                                             const int* prefetch_from, int& prefetched) {
  typedef Lane<E, VEC> Ln;
  int idx[UB];
  typename Ln::raw_t v[UB];
#pragma unroll
  for (int u = 0; u < UB; ++u) idx[u] = __shfl(myidx, src0 + u, 64);
  // the next chunk's indices are requested here -- after this chunk's were broadcast, ahead of its row
  // loads -- so they arrive under those loads and a chunk costs one round trip, not two
  if (prefetch_from != nullptr) prefetched = *prefetch_from;
#pragma unroll
  for (int u = 0; u < UB; ++u) v[u] = Ln::load(X + (size_t)max(idx[u], 0) * ldx + csafe);
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    if (!(cok && idx[u] >= 0)) v[u] = Ln::zero();
    Ln::add(acc, v[u]);
  }
}

// ------------------------------------------------------------------------------------------
// Sparse-row task body: the L lanes [lane & ~(L-1), +L) own one task (row or row segment)
// = CSR entries [e0, e0 + n); lane slot s covers columns pbase + s*VEC .. +VEC.  All control
// flow is wave-uniform (loop bounds come from the wave-wide maximum n); shorter tasks are
// predicated off by idx = -1, and adding the resulting 0.0f is exact.  The result goes to
// dstZ (a row of Z, element type) or dstP (a partial-sum row of the fp32 workspace).
// ------------------------------------------------------------------------------------------
template <typename E, int L, int VEC, bool WIDE, int UMAX = HCSPMM_SPARSE_U>
__device__ __forceinline__ void sparse_task(const typename E::T* __restrict__ X, typename E::T* __restrict__ dstZ,
                                            float* __restrict__ dstP, const int* __restrict__ col, int e0, int n,
                                            size_t ldx, int c0, int cend, int lane) {
  typedef Lane<E, VEC> Ln;
  typedef typename Ln::acc_t acc_t;
  constexpr int U = (L < UMAX) ? L : UMAX;  // loads in flight per lane
  // WIDE: the whole wave owns ONE task (e0, n wave-uniform); per 64-entry super-chunk lane i holds
  // entry base+i, so lane group g sums entries [base + g*L, base + (g+1)*L) and the 64/L group sums
  // are combined by a fixed xor-shuffle tree at the end.  Otherwise each lane group owns its own task.
  constexpr int STRIDE = WIDE ? 64 : L;
  const int s = lane & (L - 1);
  const int pos = WIDE ? lane : s;
  const int gbase = lane & ~(L - 1);
  int nmax = n;
  if (!WIDE) {
#pragma unroll
    for (int off = L; off < 64; off <<= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
  }
  nmax = __builtin_amdgcn_readfirstlane(nmax);

  for (int pbase = c0; pbase < cend; pbase += L * VEC) {  // feature columns [c0, cend) of the rows
    const bool cok = pbase + s * VEC < cend;
    const int c = cok ? lane_col<VEC>(pbase + s * VEC, cend) : 0;
    const int csafe = c;
    acc_t acc = azero<VEC>();
    int next = (pos < n) ? col[e0 + pos] : -1;
    for (int base = 0; base < nmax; base += STRIDE) {
      const int myidx = next;
      const bool more = base + STRIDE + pos < n;
      next = -1;
      const int cnt = min(L, nmax - base);  // longest lane group's share of this chunk
      const int* pf = more ? col + e0 + base + STRIDE + pos : nullptr;  // consumed by the chunk's first batch
      for (int j = 0; j < cnt;) {
        const int left = cnt - j;  // wave-uniform: short tasks (the bulk of a low-degree graph) get
        if (left > U / 2) {        // short batches instead of a full one padded with dummy loads
          gather_batch<E, VEC, U>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
          j += U;
        } else if (U >= 8 && left > U / 4) {
          gather_batch<E, VEC, (U >= 8 ? U / 2 : 1)>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
          j += U / 2;
        } else if (U >= 4 && left > 1) {
          gather_batch<E, VEC, (U >= 8 ? U / 4 : 2)>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
          j += (U >= 8 ? U / 4 : 2);
        } else {
          gather_batch<E, VEC, 1>(X, ldx, csafe, cok, myidx, gbase + j, acc, pf, next);
          j += 1;
        }
        pf = nullptr;
      }
    }
    if (WIDE) {
#pragma unroll
      for (int off = L; off < 64; off <<= 1) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) aset(acc, q, aget(acc, q) + __shfl_xor(aget(acc, q), off, 64));
      }
    }
    if (cok && (!WIDE || lane < L)) {
      if (dstZ != nullptr) Ln::store(dstZ + c, acc);
      else if (dstP != nullptr) Ln::store_partial(dstP + c, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Tiny tasks (at most two entries, indices inline in the descriptor: hcspmm.h n_tiny).  On a low-degree
// graph these are most of the rows, and a wave that handles 64/L of them is a chain of dependent round
// trips (descriptor -> indices -> rows -> store) with one or two loads in flight per lane: latency, not
// bandwidth, sets the time (tools/lowdeg_breakdown.py).  Here each lane group takes T tasks at once: T
// independent descriptor loads, then up to 2T independent row loads, then T stores -- two round trips
// per T tasks.  The sum is 0 + x[index0] + x[index1] in that order: the sequential CSR order.
// ------------------------------------------------------------------------------------------
template <typename E, int L, int VEC, int T>
__device__ __forceinline__ void tiny_tasks(const PlanArgs& a, int first, int c0, int cend, int lane) {
  typedef Lane<E, VEC> Ln;
  typedef typename E::T elem_t;
  const elem_t* X = reinterpret_cast<const elem_t*>(a.X);
  elem_t* Z = reinterpret_cast<elem_t*>(a.Z);
  constexpr int R = 64 / L;
  const int g = lane / L, s = lane & (L - 1);
  const int4* tasks = reinterpret_cast<const int4*>(a.plan + a.off_tasks);
  int4 d[T];
  bool any1 = false, any2 = false;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int tid = first + t * R + g;  // consecutive tasks (= ascending rows) across the lane groups
    d[t] = (tid < a.n_tasks) ? tasks[tid] : int4{0, -1, -1, -1};  // .z < 0: no task
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    any1 |= d[t].y >= 0;
    any2 |= d[t].w >= 0;
  }
  any1 = __builtin_amdgcn_ballot_w64(any1) != 0;  // wave-uniform: a wave inside the 0- or 1-entry class issues
  any2 = __builtin_amdgcn_ballot_w64(any2) != 0;  // no loads for the absent entries
  for (int pbase = c0; pbase < cend; pbase += L * VEC) {
    const bool cok = pbase + s * VEC < cend;
    const int c = cok ? lane_col<VEC>(pbase + s * VEC, cend) : 0;
    const int csafe = c;
    typename Ln::raw_t v0[T], v1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) v0[t] = v1[t] = Ln::zero();
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
      typename Ln::acc_t acc = azero<VEC>();
      if (d[t].y >= 0) Ln::add(acc, v0[t]);
      if (d[t].w >= 0) Ln::add(acc, v1[t]);
      if (cok && d[t].z >= 0) {
        if (d[t].x >= 0) Ln::store(Z + (size_t)d[t].x * a.ldz + c, acc);
        else Ln::store_partial(a.partial + (size_t)(-(d[t].x + 1)) * (size_t)a.D + c, acc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Dense-tile unit: one wave = (dense window, panel of 16*VEC feature columns).
// MFMA operand maps (v_mfma_f32_16x16x4_f32): lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; accumulator register r of lane l is D[4*(l >> 4) + r][l & 15].
// Instruction q of a k-step multiplies by feature column  panel + j*VEC + q, so after the VEC
// instructions lane l holds Z[row 4*(l>>4)+r][panel + j*VEC .. +VEC) -- a contiguous vector.
// ------------------------------------------------------------------------------------------
template <typename E, int VEC>
__device__ __forceinline__ void dense_store(typename E::T* __restrict__ Z, const f32x4 (&acc)[VEC], int window, int kq,
                                            int c, int N, size_t ldz) {
  typedef Lane<E, VEC> Ln;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = window * 16 + 4 * kq + r;
    if (row < N) {
      typename Ln::acc_t o;
#pragma unroll
      for (int q = 0; q < VEC; ++q) aset(o, q, acc[q][r]);
      Ln::store(Z + (size_t)row * ldz + c, o);
    }
  }
}

// TR = false: acc holds the 16 x (16*VEC) tile in the MFMA C layout described above (dense_store writes it).
// TR = true (fused aggregate+update): the operands are exchanged -- A operand = the gathered X values, B operand =
// the 0/1 tile -- so the matrix core produces the TRANSPOSED tile: lane l, register r of acc[q] holds
// Z[row l & 15][panel + (4*(l >> 4) + r)*VEC + q].  Same products, same k order => the same bits; the point is that
// a lane now holds ONE row's values, indexed by l >> 4, which is exactly the A-operand shape of the following
// (tile x weights) MFMAs -- the tile never leaves the registers (fused_epilogue).
template <bool TR>
__device__ __forceinline__ f32x4 tile_mfma(float a01, float x, f32x4 acc) {
  if constexpr (TR) return __builtin_amdgcn_mfma_f32_16x16x4f32(x, a01, acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x4f32(a01, x, acc, 0, 0, 0);
}

// DB: k-steps per batch at 16 bytes per lane (a template parameter, not the macro itself: fused_rows.hip runs this chain with
// a smaller batch, and one template with two bodies in two translation units would be an ODR violation)
template <typename E, int VEC, bool TR, int DB>
__device__ __forceinline__ void dense_chain(const typename E::T* __restrict__ X, const int* __restrict__ U, cu64_p masks,
                                            int K4, int csafe, bool cok, size_t ldx, int lane, f32x4 (&acc)[VEC]) {
  typedef Lane<E, VEC> Ln;
  const int kq = lane >> 4;
  for (int kb = 0; kb < K4; kb += 16) {
    const int myU = (kb * 4 + lane < K4 * 4) ? U[kb * 4 + lane] : -1;
    const int steps = min(16, K4 - kb);
    // same bytes in flight per lane whatever the panel width
    constexpr int B = DB * 16 / (VEC * (int)sizeof(typename E::T));
    for (int t0 = 0; t0 < steps; t0 += B) {
      // Branch-free batch (see sparse_task): padded columns (U = -1) and steps past the end re-read
      // row 0 and are zeroed by the select; their A tile is zero as well.
      int idx[B];
      typename Ln::raw_t x[B];
      float a[B];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int t = t0 + u;
        idx[u] = __shfl(myU, (4 * t + kq) & 63, 64);
        const unsigned long long m = masks[min(kb + t, K4 - 1)];  // wave-uniform address: scalar load
        a[u] = (t < steps && ((m >> lane) & 1ull)) ? 1.0f : 0.0f;
        if (t >= steps) idx[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < B; ++u) x[u] = Ln::load(X + (size_t)max(idx[u], 0) * ldx + csafe);
#pragma unroll
      for (int u = 0; u < B; ++u) {
        if (!(cok && idx[u] >= 0)) x[u] = Ln::zero();
        if (t0 + u < steps) {  // wave-uniform: no MFMA issue slots for the padding of a short batch
#pragma unroll
          for (int q = 0; q < VEC; ++q) acc[q] = tile_mfma<TR>(a[u], Ln::elem(x[u], q), acc[q]);
        }
      }
    }
  }
}

template <typename E, int VEC>
__device__ __forceinline__ void dense_unit(const typename E::T* __restrict__ X, typename E::T* __restrict__ Z,
                                           const int* __restrict__ U, cu64_p masks, int K4, int window, int panel,
                                           int N, int D, size_t ldx, size_t ldz, int lane) {
  const int kq = lane >> 4, j = lane & 15;
  const bool cok = panel * 16 * VEC + j * VEC < D;
  const int c = cok ? lane_col<VEC>(panel * 16 * VEC + j * VEC, D) : 0;
  const int csafe = c;
  f32x4 acc[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  dense_chain<E, VEC, false, HCSPMM_DENSE_B>(X, U, masks, K4, csafe, cok, ldx, lane, acc);
  if (cok) dense_store<E, VEC>(Z, acc, window, kq, c, N, ldz);
}

// Compact dense unit: a window whose whole description is one fixed-size record at an address that follows
// from the unit number (hcspmm.h n_dense_compact / n_dense_compact2): C words per lane -- lane l loads words
// l, 64 + l, ... with coalesced 256-byte loads -- and the wave can gather: window / K4 / masks go to scalar
// registers (v_readlane), the column list is broadcast as in dense_unit.  Two round trips per unit instead of
// three.  C = 1: K <= 40 (64-word record); C = 2: K <= 80 (128 words).  Layout: [window, K/4, U[KMAX],
// KMAX/4 x (mask lo, mask hi)].  Same MFMA chain as dense_unit, so the same bits.
template <int C> struct Rec {
  int w[C];
  // word `i` of the record (compile-time i): a scalar
  template <int I> __device__ __forceinline__ int scalar() const { return __builtin_amdgcn_readlane(w[I / 64], I % 64); }
  // words i0 + kq (kq = 0..3 per lane): a broadcast; the four indices may straddle two of the lane words
  template <int I0> __device__ __forceinline__ int gather4(int kq) const {
    if constexpr (I0 / 64 == (I0 + 3) / 64) {
      return __shfl(w[I0 / 64], (I0 + kq) & 63, 64);
    } else {
      const int lo = __shfl(w[I0 / 64], (I0 + kq) & 63, 64);
      const int hi = __shfl(w[(I0 + 3) / 64], (I0 + kq) & 63, 64);
      return (I0 + kq) / 64 == I0 / 64 ? lo : hi;
    }
  }
};

template <typename E, int VEC, int C, int KMAX, int STEPS, int T0, bool TR = false>
struct CompactSteps {
  template <int U> static __device__ __forceinline__ void meta(const Rec<C>& rec, int K4, int kq, int lane, int* idx, float* a) {
    if constexpr (U < STEPS) {
      constexpr int t = T0 + U;
      idx[U] = rec.template gather4<2 + 4 * t>(kq);
      const unsigned lo = (unsigned)rec.template scalar<2 + KMAX + 2 * t>();
      const unsigned hi = (unsigned)rec.template scalar<3 + KMAX + 2 * t>();
      const unsigned long long m = ((unsigned long long)hi << 32) | lo;
      a[U] = (t < K4 && ((m >> lane) & 1ull)) ? 1.0f : 0.0f;
      if (t >= K4) idx[U] = -1;
      meta<U + 1>(rec, K4, kq, lane, idx, a);
    }
  }
  static __device__ __forceinline__ void run(const typename E::T* __restrict__ X, const Rec<C>& rec, int K4, int csafe,
                                             bool cok, size_t ldx, int lane, f32x4 (&acc)[VEC]) {
    typedef Lane<E, VEC> Ln;
    int idx[STEPS];
    typename Ln::raw_t x[STEPS];
    float a[STEPS];
    meta<0>(rec, K4, lane >> 4, lane, idx, a);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) x[u] = Ln::load(X + (size_t)max(idx[u], 0) * ldx + csafe);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      if (!(cok && idx[u] >= 0)) x[u] = Ln::zero();
      if (T0 + u < K4) {  // wave-uniform
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = tile_mfma<TR>(a[u], Ln::elem(x[u], q), acc[q]);
      }
    }
  }
};

// the MFMA chain of a compact record (K4 k-steps), any operand order
template <typename E, int VEC, int C, bool TR>
__device__ __forceinline__ void compact_chain(const typename E::T* __restrict__ X, const Rec<C>& rec, int K4, int csafe,
                                              bool cok, size_t ldx, int lane, f32x4 (&acc)[VEC]) {
  constexpr int KMAX = C == 1 ? HCSPMM_COMPACT_K : HCSPMM_COMPACT2_K;
  if constexpr (C == 1) {  // K4 <= 10
    if (K4 <= 2) CompactSteps<E, VEC, C, KMAX, 2, 0, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
    else if (K4 <= 4) CompactSteps<E, VEC, C, KMAX, 4, 0, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
    else {
      CompactSteps<E, VEC, C, KMAX, 8, 0, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
      if (K4 > 8) CompactSteps<E, VEC, C, KMAX, 2, 8, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);  // K = 40
    }
  } else {  // 12 <= K4 <= 20
    CompactSteps<E, VEC, C, KMAX, 8, 0, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
    if (K4 <= 12) CompactSteps<E, VEC, C, KMAX, 4, 8, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
    else {
      CompactSteps<E, VEC, C, KMAX, 8, 8, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
      if (K4 > 16) CompactSteps<E, VEC, C, KMAX, 4, 16, TR>::run(X, rec, K4, csafe, cok, ldx, lane, acc);
    }
  }
}

template <typename E, int VEC, int C>
__device__ __forceinline__ void dense_compact_unit(const typename E::T* __restrict__ X, typename E::T* __restrict__ Z,
                                                   const int* __restrict__ recp, int panel, int N, int D, size_t ldx,
                                                   size_t ldz, int lane) {
  constexpr int KMAX = C == 1 ? HCSPMM_COMPACT_K : HCSPMM_COMPACT2_K;
  static_assert(HCSPMM_COMPACT_K == 40 && HCSPMM_COMPACT2_K == 80 && 2 + KMAX + KMAX / 2 <= 64 * C, "compact record layout");
  Rec<C> rec;
#pragma unroll
  for (int c = 0; c < C; ++c) rec.w[c] = recp[64 * c + lane];
  const int window = rec.template scalar<0>();
  const int K4 = rec.template scalar<1>();
  const int kq = lane >> 4, j = lane & 15;
  const bool cok = panel * 16 * VEC + j * VEC < D;
  const int c = cok ? lane_col<VEC>(panel * 16 * VEC + j * VEC, D) : 0;
  const int csafe = c;
  f32x4 acc[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  compact_chain<E, VEC, C, false>(X, rec, K4, csafe, cok, ldx, lane, acc);
  if (cok) dense_store<E, VEC>(Z, acc, window, kq, c, N, ldz);
}

// ------------------------------------------------------------------------------------------
// Fused aggregate + update for dense-tile windows (hcspmm_forward_fused; replaces the second stage of the
// reference's fused kernels, hybrid_all_kernel.cu:1807-1837 and its siblings, which keep the aggregated tile in
// shared memory as WMMA A-operand blocks).  Here the tile never leaves the registers: the aggregation runs with
// exchanged MFMA operands (TR = true above), so lane l, register r of acc[q] holds tile[row l & 15][feature
// f(l >> 4, r, q)], f = panel + (4*(l >> 4) + r)*DV + q -- one row per lane, the feature indexed by l >> 4: the
// A-operand shape.  The update is then 4*DV MFMA steps per 16-column output tile, step (r, q) contracting over
// the four features f(0..3, r, q) with B operand W[f(l >> 4, r, q)][16*t + (l & 15)] read from LDS (W is staged
// once per workgroup, zero-padded to whole panels; the row stride HS makes the four l >> 4 groups hit disjoint
// banks).  One wave owns a WINDOW (all its column panels in turn, out accumulated across them); out2 = A*X is
// written from the transposed layout (each lane stores 4*DV consecutive floats of its row).  The contraction
// order over the D features is a fixed permutation => deterministic; out2 has the unfused kernel's bits.
// ------------------------------------------------------------------------------------------

__host__ __device__ inline int fused_row_stride(int H, int dv) { return H + (dv >= 4 ? 1 : (dv == 2 ? 2 : 4)); }

template <int DV, int HT>  // H == 16*HT exactly: no guards, so the whole update of a panel is one basic block
__device__ __forceinline__ void fused_epilogue(const f32x4 (&acc)[DV], const float* __restrict__ s_w, int HS, int panel,
                                               int lane, f32x4 (&oacc)[HT]) {
  const int kq = lane >> 4, n = lane & 15;
  const float* wbase = s_w + (panel * 16 * DV + 4 * kq * DV) * HS + n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float w[DV][HT];  // the LDS reads of a step group are issued together, ahead of its MFMAs
#pragma unroll
    for (int q = 0; q < DV; ++q)
#pragma unroll
      for (int t = 0; t < HT; ++t) w[q][t] = wbase[(r * DV + q) * HS + 16 * t];
#pragma unroll
    for (int q = 0; q < DV; ++q)
#pragma unroll
      for (int t = 0; t < HT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[q][r], w[q][t], oacc[t], 0, 0, 0);
  }
}

template <int DV>
__device__ __forceinline__ void store_tile_transposed(float* __restrict__ Z, const f32x4 (&acc)[DV], int window, int panel,
                                                      int N, int D, size_t ldz, int lane) {
  typedef Lane<F32, DV> Ln;
  const int kq = lane >> 4, row = window * 16 + (lane & 15);
  if (row >= N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = panel * 16 * DV + (4 * kq + r) * DV;
    if (c < D) {
      typename Ln::acc_t o;
#pragma unroll
      for (int q = 0; q < DV; ++q) aset(o, q, acc[q][r]);
      // a lane's four stores together cover 16*DV consecutive bytes, but one store instruction writes 16-byte
      // pieces 64*DV bytes apart: plain (cached) stores, so that the L2 assembles whole lines before they leave
      *reinterpret_cast<typename Ln::acc_t*>(Z + (size_t)row * ldz + c) = o;
    }
  }
}

template <int DV, int HT>  // HT = H / 16 output tiles, held in registers
__device__ __forceinline__ void fused_dense_window(const PlanArgs& a, int unit, const float* __restrict__ s_w, int HS,
                                                   int lane) {
  const float* X = reinterpret_cast<const float*>(a.X);
  float* Z = reinterpret_cast<float*>(a.Z);
  const int kq = lane >> 4, j = lane & 15;
  const int n_reg = a.n_dense - a.n_dense_compact - a.n_dense_compact2;
  f32x4 oacc[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int window;
  // one panel: aggregate (transposed tile in acc), write out2, chain the update MFMAs
#define HCSPMM_FUSED_PANELS(CHAIN)                                                        \
  for (int panel = 0; panel < a.n_panels; ++panel) {                                      \
    const int c = panel * 16 * DV + j * DV;                                               \
    const bool cok = c < a.D;                                                             \
    const int csafe = cok ? c : 0;                                                        \
    f32x4 acc[DV];                                                                        \
    _Pragma("unroll") for (int q = 0; q < DV; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};    \
    CHAIN;                                                                                \
    store_tile_transposed<DV>(Z, acc, window, panel, a.N, a.D, a.ldz, lane);              \
    fused_epilogue<DV, HT>(acc, s_w, HS, panel, lane, oacc);                              \
  }
  if (unit < n_reg) {
    cint_p dix = (cint_p)(a.plan + a.off_dense_index) + 4 * unit;  // wave-uniform: scalar loads
    window = dix[0];
    const int K4 = dix[2];
    const int* U = a.plan + a.off_dense_pack + dix[1];
    cu64_p masks = (cu64_p)(U + 4 * K4);
    HCSPMM_FUSED_PANELS((dense_chain<F32, DV, true, HCSPMM_DENSE_B>(X, U, masks, K4, csafe, cok, a.ldx, lane, acc)))
  } else if (unit < n_reg + a.n_dense_compact2) {
    Rec<2> rec;
    const int* recp = a.plan + a.off_dense_compact2 + (unit - n_reg) * HCSPMM_COMPACT2_WORDS;
    rec.w[0] = recp[lane];
    rec.w[1] = recp[64 + lane];
    window = rec.template scalar<0>();
    const int K4 = rec.template scalar<1>();
    HCSPMM_FUSED_PANELS((compact_chain<F32, DV, 2, true>(X, rec, K4, csafe, cok, a.ldx, lane, acc)))
  } else {
    Rec<1> rec;
    rec.w[0] = (a.plan + a.off_dense_compact + (unit - n_reg - a.n_dense_compact2) * HCSPMM_COMPACT_WORDS)[lane];
    window = rec.template scalar<0>();
    const int K4 = rec.template scalar<1>();
    HCSPMM_FUSED_PANELS((compact_chain<F32, DV, 1, true>(X, rec, K4, csafe, cok, a.ldx, lane, acc)))
  }
#undef HCSPMM_FUSED_PANELS
  // out: accumulator register r of lane (kq, j) is out[row 4*kq + r][16*t + j]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = window * 16 + 4 * kq + r;
    if (row < a.N) {
#pragma unroll
      for (int t = 0; t < HT; ++t) a.out[(size_t)row * (size_t)(16 * HT) + 16 * t + j] = oacc[t][r];
    }
  }
}

// the dense region of a fused launch: stage W in LDS (every wave of the workgroup takes part), then the workgroup's
// waves stride over the dense windows, one window per wave at a time
template <int DV, int HT>
__device__ __forceinline__ void fused_dense_loop(const PlanArgs& a, const float* __restrict__ s_w, int HS, int lane, int wave) {
  for (int unit = ((int)blockIdx.x - a.sparse_wgs) * kWaves + wave; unit < a.n_dense; unit += a.fused_dense_wgs * kWaves)
    fused_dense_window<DV, HT>(a, unit, s_w, HS, lane);
}

__device__ __forceinline__ void fused_dense_region(const PlanArgs& a, int lane, int wave) {
  extern __shared__ __attribute__((aligned(16))) float s_fused_w[];
  const int dv = a.dense_vec;
  const int HS = fused_row_stride(a.H, dv);
  const int rows = a.n_panels * 16 * dv;  // D rounded up to whole panels; the padding rows are zero
  for (int i = threadIdx.x; i < rows * a.H; i += kThreads) {
    const int k = i / a.H, h = i - k * a.H;
    s_fused_w[k * HS + h] = k < a.D ? a.W[(long long)k * a.w_ldr + (long long)h * a.w_ldc] : 0.0f;
  }
  __syncthreads();
  // W is staged once per workgroup, so the region is a bounded number of workgroups that stride over the windows
  if (dv == 4) a.H == 32 ? fused_dense_loop<4, 2>(a, s_fused_w, HS, lane, wave) : fused_dense_loop<4, 1>(a, s_fused_w, HS, lane, wave);
  else a.H == 32 ? fused_dense_loop<2, 2>(a, s_fused_w, HS, lane, wave) : fused_dense_loop<2, 1>(a, s_fused_w, HS, lane, wave);
}

// ------------------------------------------------------------------------------------------
// Planned hybrid kernel: ONE launch covers both sub-paths (as the reference's single launch
// does, K.cu:960/1039) -- per column panel, workgroups [0, wide_wgs) run wide sparse tasks, then ordinary
// ones, then tiny ones (the last tiny_wgs); after all sparse panels come the dense units.
// ------------------------------------------------------------------------------------------
// UNROLL row loads in flight per lane, MINW waves per SIMD the register budget must allow.  With the
// branch-free batches <8, 4> is best for throughput- and latency-bound launches alike
// (profiles/r01/ab_u_b_mw_v2.log; before them <4, 8> won: profiles/r01/ab_u_b_mw.log).
// (positions among the free workgroups of a panel, i.e. behind the slice_wgs XCD-bound ones; pad_wgs idle ones close the panel)
__device__ __forceinline__ int sparse_wgs_pp_ordinary_end(const PlanArgs& a) { return a.free_wgs_pp - a.tiny_wgs; }

// second-widest and narrowest dense-tile vector of a VEC-wide build
template <int VEC> struct DenseV {
  static constexpr int mid = VEC >= 8 ? 4 : (VEC >= 4 ? 2 : 1);  // fp32: 4 -> 2; 16-bit: 8 -> 4
};

template <typename E, int L, int VEC, int UNROLL, int MINW, bool FUSED = false>
__global__ __launch_bounds__(kThreads, MINW) void hybrid_plan_kernel(PlanArgs a) {
  typedef typename E::T elem_t;
  const elem_t* X = reinterpret_cast<const elem_t*>(a.X);
  elem_t* Z = reinterpret_cast<elem_t*>(a.Z);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((int)blockIdx.x < a.sparse_wgs) {
    // column-panel-major: every sparse task runs once per panel of a.panel_cols feature columns, and
    // all workgroups of panel p precede those of panel p+1, so at any time the gathers touch one
    // 128-byte slice of the X rows -- four times as many distinct rows fit the per-XCD L2
    const int p = (int)blockIdx.x / a.sparse_wgs_pp;
    const int b = (int)blockIdx.x - p * a.sparse_wgs_pp;
    const int c0 = p * a.panel_cols;
    const int cend = min(a.D, c0 + a.panel_cols);
    // position among the free (not XCD-bound) workgroups of the panel; < 0: one of the XCD-bound ones in front of them.
    // (Alternating groups of sliced and free workgroups, so that L2-bound and fabric-bound work overlap, was built and
    // measured: 0-4 % slower everywhere -- the free tasks' lines evict the slice's: profiles/r03/ab_slice_interleave.log.)
    const int bf = b - a.slice_wgs;
    if (bf >= 0 && bf < a.wide_wgs) {
      // wide tasks: the a.n_wide longest tasks, one per wave
      const int tid = bf * kWaves + wave;
      if (tid >= a.n_wide) return;
      const int4 t = reinterpret_cast<const int4*>(a.plan + a.off_tasks)[tid];
      elem_t* dz = (t.w < 0) ? Z + (size_t)t.x * a.ldz : nullptr;
      float* dp = (t.w < 0) ? nullptr : a.partial + (size_t)t.w * (size_t)a.D;
      sparse_task<E, L, VEC, true, UNROLL>(X, dz, dp, a.col, __builtin_amdgcn_readfirstlane(t.y),
                                           __builtin_amdgcn_readfirstlane(t.z), a.ldx, c0, cend, lane);
    } else if (bf >= sparse_wgs_pp_ordinary_end(a)) {
      // (bf >= free_wgs_pp: the workgroups that pad a sliced panel to a multiple of 8 -- with the tiny tasks in their own
      // launch, tiny_wgs = 0, they would otherwise sum real tiny tasks a second time: same values, wasted work)
      if (bf >= a.free_wgs_pp) return;
      constexpr int R = 64 / L;
      const int first = a.n_tasks - a.n_tiny + ((bf - sparse_wgs_pp_ordinary_end(a)) * kWaves + wave) * (R * TinyT<L>::value);
      if (first >= a.n_tasks) return;
      tiny_tasks<E, L, VEC, TinyT<L>::value>(a, first, c0, cend, lane);
    } else {
      // ordinary tasks, and the XCD-affine sliced ones: R per wave, one lane group each, strictly in CSR order
      constexpr int R = 64 / L;
      const int g = lane / L;
      const int4* tp = nullptr;
      if (bf < 0) {
        // sliced region: workgroup b serves the slices s = b (mod 8) -- all workgroups that share its XCD's L2 gather from
        // the same 1/8 of the X rows.  The slice lists are padded to whole waves (64 descriptors), so a wave never straddles
        // two lists; padding descriptors have row -1.
        cint_p tbl = (cint_p)(a.plan + a.off_slice_table);  // wave-uniform: scalar loads
        int j = ((b >> 3) * kWaves + wave) * R;             // first descriptor of this wave in its XCD's concatenated lists
        for (int sl = b & 7; sl < a.n_slices; sl += 8) {
          const int lo = tbl[sl], cnt = tbl[sl + 1] - lo;
          if (j < cnt) {
            tp = reinterpret_cast<const int4*>(a.plan + a.off_slice_tasks) + lo + j + g;
            break;
          }
          j -= cnt;
        }
      } else {
        const int tid = a.n_wide + ((bf - a.wide_wgs) * kWaves + wave) * R + g;
        if (tid < a.n_tasks - a.n_tiny) tp = reinterpret_cast<const int4*>(a.plan + a.off_tasks) + tid;
      }
      int e0 = 0, n = 0;
      elem_t* dz = nullptr;
      float* dp = nullptr;
      if (tp != nullptr) {
        const int4 t = *tp;
        if (t.x >= 0) {
          e0 = t.y;
          n = t.z;
          if (t.w < 0) dz = Z + (size_t)t.x * a.ldz;
          else dp = a.partial + (size_t)t.w * (size_t)a.D;
        }
      }
      sparse_task<E, L, VEC, false, UNROLL>(X, dz, dp, a.col, e0, n, a.ldx, c0, cend, lane);
    }
  } else {
    if constexpr (FUSED) {  // fp32 only: dense windows aggregate AND multiply by the weights (one window per wave)
      fused_dense_region(a, lane, wave);
      return;
    }
    constexpr int VM = DenseV<VEC>::mid;
    int unit = ((int)blockIdx.x - a.sparse_wgs) * kWaves + wave;
    if (unit >= a.n_dense * a.n_panels) return;
    // regular windows (K > 80) first, widest first; then the double-record ones (K <= 80), then the compact ones (K <= 40)
    const int n_reg = a.n_dense - a.n_dense_compact - a.n_dense_compact2;
    if (unit >= n_reg * a.n_panels) {
      unit -= n_reg * a.n_panels;
      // elements per lane on the dense-tile path (a panel is 16*dense_vec columns): set by the launcher from D
      if (unit < a.n_dense_compact2 * a.n_panels) {
        const int panel = unit / a.n_dense_compact2, ci = unit - panel * a.n_dense_compact2;  // panel-major
        const int* rec = a.plan + a.off_dense_compact2 + ci * HCSPMM_COMPACT2_WORDS;
        if (a.dense_vec == VEC) dense_compact_unit<E, VEC, 2>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
        else if (a.dense_vec == VM) dense_compact_unit<E, VM, 2>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
        else dense_compact_unit<E, 1, 2>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
        return;
      }
      unit -= a.n_dense_compact2 * a.n_panels;
      const int panel = unit / a.n_dense_compact, ci = unit - panel * a.n_dense_compact;  // panel-major
      const int* rec = a.plan + a.off_dense_compact + ci * HCSPMM_COMPACT_WORDS;
      if (a.dense_vec == VEC) dense_compact_unit<E, VEC, 1>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
      else if (a.dense_vec == VM) dense_compact_unit<E, VM, 1>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
      else dense_compact_unit<E, 1, 1>(X, Z, rec, panel, a.N, a.D, a.ldx, a.ldz, lane);
      return;
    }
    const int panel = unit / n_reg, di = unit - panel * n_reg;  // panel-major, like the sparse region
    cint_p dix = (cint_p)(a.plan + a.off_dense_index) + 4 * di;  // wave-uniform: scalar loads
    const int4 d = int4{dix[0], dix[1], dix[2], dix[3]};
    const int* U = a.plan + a.off_dense_pack + d.y;
    cu64_p masks = (cu64_p)(U + 4 * d.z);
    if (a.dense_vec == VEC) dense_unit<E, VEC>(X, Z, U, masks, d.z, d.x, panel, a.N, a.D, a.ldx, a.ldz, lane);
    else if (a.dense_vec == VM) dense_unit<E, VM>(X, Z, U, masks, d.z, d.x, panel, a.N, a.D, a.ldx, a.ldz, lane);
    else dense_unit<E, 1>(X, Z, U, masks, d.z, d.x, panel, a.N, a.D, a.ldx, a.ldz, lane);
  }
}

// Tiny tasks as their own launch (low-degree graphs: most rows of the paper's Table-II shapes hold at most two entries).
// Inside the hybrid kernel the tiny region shares that kernel's register allocation (96 registers, five waves per SIMD);
// this kernel holds nothing but the tiny-task state -- 56-64 registers, eight waves per SIMD.  Measured
// (profiles/r03/ab_tiny_kernel.log): 2-4 % on the RD / TT / YeastH-sized graphs with T = 2 tasks per lane group; T = 4
// is no better and T = 8 (96 registers, five waves) 25 % worse -- the launch is NOT short of loads in flight.  What it is
// bound by (profiles/r03/lowdeg_floor.log, RD-sized, D = 32): descriptors + Z stores alone 104 us (a Z-sized fill: 89),
// + every gather an L2 hit 183 us, + the real 622 MB X 321 us = 1.65 GB at 5.1 TB/s of random 128-byte lines mixed with
// stores, 0.8 of what the guide calls achievable for STREAMING.  Bytes are within 1.2x of compulsory: nothing left to fuse.
// Taken when the plan has enough tiny tasks to fill the chip (launch_plan_LV); small graphs keep the in-kernel region
// and its single launch.  Same order of additions: same bits.
#ifndef HCSPMM_TINY_KERNEL_T
#define HCSPMM_TINY_KERNEL_T 2
#endif
#ifndef HCSPMM_TINY_KERNEL_WAVES
#define HCSPMM_TINY_KERNEL_WAVES 8
#endif
#ifndef HCSPMM_TINY_KERNEL_MIN_TASKS
#define HCSPMM_TINY_KERNEL_MIN_TASKS 524288  // fewer tiny tasks than this stay in the hybrid launch: even at 465 K, -15 % at 240 K, +3-4 % at 900 K (profiles/r03/ab_tiny_sizes.log)
#endif
template <typename E, int L, int VEC>
__global__ __launch_bounds__(kThreads, HCSPMM_TINY_KERNEL_WAVES) void tiny_kernel(PlanArgs a) {
  constexpr int R = 64 / L, T = HCSPMM_TINY_KERNEL_T;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = (int)blockIdx.x / a.tiny_kernel_wgs;
  const int b = (int)blockIdx.x - p * a.tiny_kernel_wgs;
  const int c0 = p * a.panel_cols;
  const int first = a.n_tasks - a.n_tiny + (b * kWaves + wave) * (R * T);
  if (first >= a.n_tasks) return;
  tiny_tasks<E, L, VEC, T>(a, first, c0, min(a.D, c0 + a.panel_cols), lane);
}

// Fix-up: rows that were split into segments -- Z[row] = sum of its partial rows, in a fixed order
// (deterministic): one wave per split row; its 64/L lane groups (L = lanes a row needs) take every (64/L)-th
// segment with eight loads in flight and are combined by the same xor-shuffle tree as a wide task.  A hub of
// hundreds of thousands of entries has more than a thousand segments: summed by one lane group, four loads at a
// time, its wave alone took longer than the rest of the fix-up launch.  Partials are fp32 whatever the feature type.
template <typename E, int VEC>
__global__ __launch_bounds__(kThreads) void fixup_kernel(PlanArgs a) {
  typedef Lane<E, VEC> Ln;
  typedef typename Ln::acc_t acc_t;
  typename E::T* Z = reinterpret_cast<typename E::T*>(a.Z);
  const int lane = threadIdx.x & 63;
  const int fi = (int)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (fi >= a.n_split_rows) return;
  const int4 f = reinterpret_cast<const int4*>(a.plan + a.off_fixups)[fi];
  const int row = f.x, s0 = f.y, ns = f.z;
  const int slots = (a.D + VEC - 1) / VEC;
  int L = 1;
  while (L < slots && L < 64) L <<= 1;  // lanes per row (a.D > 64*VEC: several column passes of 64 lanes)
  const int R = 64 / L, g = lane / L, sl = lane & (L - 1);
  for (int c0 = 0; c0 < a.D; c0 += L * VEC) {
    const bool cok = c0 + sl * VEC < a.D;
    const int c = cok ? lane_col<VEC>(c0 + sl * VEC, a.D) : 0;
    const float* p = a.partial + (size_t)s0 * (size_t)a.D + c;
    acc_t acc = azero<VEC>();
    int s = g;
    for (; s + 7 * R < ns; s += 8 * R) {  // eight independent loads in flight per lane
      acc_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = Lane<F32, VEC>::load_partial(p + (size_t)(s + u * R) * (size_t)a.D);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < ns; s += R) acc += Lane<F32, VEC>::load_partial(p + (size_t)s * (size_t)a.D);
    for (int off = L; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < VEC; ++q) aset(acc, q, aget(acc, q) + __shfl_xor(aget(acc, q), off, 64));
    }
    if (cok && g == 0) Ln::store(Z + (size_t)row * a.ldz + c, acc);
  }
}

// ------------------------------------------------------------------------------------------
// Plan-free kernel (callers that pass the reference's [0] placeholders): one workgroup per
// 16-row window, branch on hybrid_type[window] exactly like the reference launch.  Sparse
// windows deal their rows to the workgroup's lane groups; dense windows rebuild the window's
// unique-column list and tile masks in LDS from edgeToColumn / edgeToRow / column_index (the
// reference builds sparse_A / sparse_AToX_index the same way, K.cu:1067-1074) in chunks of
// kChunkK condensed columns, so any blockPartition is handled.
// ------------------------------------------------------------------------------------------
constexpr int kChunkK = 512;  // condensed columns per LDS pass (128 k-steps)
constexpr int kPlanFreeWide = 64;  // plan-free kernel: rows longer than this are summed by a whole wave

template <typename E, int L, int VEC>
__global__ __launch_bounds__(kThreads) void hybrid_window_kernel(WindowArgs a) {
  typedef Lane<E, VEC> Ln;
  typedef typename E::T elem_t;
  const elem_t* X = reinterpret_cast<const elem_t*>(a.X);
  elem_t* Z = reinterpret_cast<elem_t*>(a.Z);
  __shared__ int s_U[kChunkK];
  __shared__ unsigned int s_mask[kChunkK / 4 * 2];  // 64-bit lane masks as two 32-bit halves
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nthreads = (int)blockDim.x, nwaves = nthreads >> 6;  // 1-4 waves: sized to the window's work
  const int w = blockIdx.x;
  const int r0 = w * 16, r1 = min(r0 + 16, a.N);
  if (a.hybrid_type[w] == 0) {
    constexpr int R = 64 / L;
    const int G = R * nwaves;  // lane groups per workgroup
    const int gi = wave * R + lane / L;
    // rows up to kPlanFreeWide entries: one lane group each, strict CSR order
    for (int rb = r0; rb < r1; rb += G) {  // uniform
      const int r = rb + gi;
      int e0 = 0, n = 0;
      elem_t* dst = nullptr;
      if (r < r1) {
        e0 = a.rowptr[r];
        n = a.rowptr[r + 1] - e0;
        dst = Z + (size_t)r * a.ldz;
        if (R > 1 && n > kPlanFreeWide) {  // left to the whole-wave pass below
          n = 0;
          dst = nullptr;
        }
      }
      sparse_task<E, L, VEC, false>(X, dst, nullptr, a.col, e0, n, a.ldx, 0, a.D, lane);
    }
    // longer rows: whole waves (all 64/L lane groups on one row, shuffle-tree combine), dealt
    // round-robin over the workgroup's waves -- a hub row no longer crawls on one lane group
    if (R > 1) {
      int k = 0;
      for (int r = r0; r < r1; ++r) {  // uniform scan of the window's rows
        const int e0 = a.rowptr[r];
        const int n = a.rowptr[r + 1] - e0;
        if (n > kPlanFreeWide) {
          if (k % nwaves == wave)
            sparse_task<E, L, VEC, true>(X, Z + (size_t)r * a.ldz, nullptr, a.col, e0, n, a.ldx, 0, a.D, lane);
          ++k;
        }
      }
    }
    return;
  }
  const int lo = a.rowptr[r0], hi = a.rowptr[r1];
  const int K = a.blockPartition[w] * 8;
  const int n_panels = (a.D + 16 * VEC - 1) / (16 * VEC);
  const int kq = lane >> 4, j = lane & 15;
  for (int pb = 0; pb < n_panels; pb += nwaves) {  // uniform over the workgroup
    const int panel = pb + wave;
    const bool cok = panel < n_panels && panel * 16 * VEC + j * VEC < a.D;
    const int c = cok ? lane_col<VEC>(panel * 16 * VEC + j * VEC, a.D) : 0;
    f32x4 acc[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += kChunkK) {  // uniform
      const int kc = min(kChunkK, K - k0);
      __syncthreads();
      for (int i = threadIdx.x; i < kChunkK; i += nthreads) s_U[i] = -1;
      for (int i = threadIdx.x; i < kChunkK / 2; i += nthreads) s_mask[i] = 0u;
      __syncthreads();
      for (int e = lo + (int)threadIdx.x; e < hi; e += nthreads) {
        const int cc = a.edgeToColumn[e] - k0;
        if (cc >= 0 && cc < kc) {
          const int rl = a.edgeToRow[e] - r0;
          const int bit = 16 * (cc & 3) + rl;  // MFMA A-operand lane of (row rl, k = cc & 3)
          atomicOr(&s_mask[(cc >> 2) * 2 + (bit >> 5)], 1u << (bit & 31));
          s_U[cc] = a.col[e];
        }
      }
      __syncthreads();
      const int steps = (kc + 3) / 4;
      for (int t0 = 0; t0 < steps; t0 += 4) {
        typename Ln::raw_t x[4];
        float av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + u;
          const bool tv = t < steps;
          const int idx = tv ? s_U[min(4 * t + kq, kChunkK - 1)] : -1;
          const unsigned int mw = tv ? s_mask[t * 2 + (lane >> 5)] : 0u;
          av[u] = ((mw >> (lane & 31)) & 1u) ? 1.0f : 0.0f;
          x[u] = Ln::zero();
          if (cok && idx >= 0) x[u] = Ln::load(X + (size_t)idx * a.ldx + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int q = 0; q < VEC; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], Ln::elem(x[u], q), acc[q], 0, 0, 0);
        }
      }
    }
    if (cok) dense_store<E, VEC>(Z, acc, w, kq, c, a.N, a.ldz);
  }
}

// ------------------------------------------------------------------------------------------
// Host-side dispatch on (L, VEC).
// ------------------------------------------------------------------------------------------
template <typename E, int L, int VEC>
static hipError_t launch_plan_LV(const PlanArgs& a, hipStream_t stream) {
  constexpr int R = 64 / L;
  PlanArgs b = a;
  b.n_wide = (R > 1) ? a.n_wide : 0;  // with one lane group per wave a wide task is an ordinary one
  // a.fused bit 0: dense-tile windows multiply their tile by the weights in this launch (FUSED kernel); bit 1: the ordinary and
  // tiny tasks of this call run in the row-tile fused launch (fused_rows.hip) -- this launch keeps the sliced region, the
  // wide tasks and the dense units
  const bool dense_fused = (a.fused & 1) != 0;
  if (a.fused & 2) {
    b.n_tasks = b.n_wide;
    b.n_tiny = 0;
    b.n_dense = b.n_dense_compact = b.n_dense_compact2 = 0;  // (dense windows are tiles of that launch as well)
  }
  b.wide_wgs = (b.n_wide + kWaves - 1) / kWaves;
  // tiny tasks: a region of the hybrid launch, or -- when there are enough of them -- a launch of their own behind it
  static const int tiny_kernel_min = [] {
    const char* e = getenv("HCSPMM_TINY_KERNEL_MIN_TASKS");
    return e ? atoi(e) : HCSPMM_TINY_KERNEL_MIN_TASKS;
  }();
  const bool own_tiny_launch = !a.fused && tiny_kernel_min >= 0 && b.n_tiny >= tiny_kernel_min && b.n_tiny > 0;
  b.tiny_kernel_wgs = own_tiny_launch ? (b.n_tiny + kWaves * R * HCSPMM_TINY_KERNEL_T - 1) / (kWaves * R * HCSPMM_TINY_KERNEL_T) : 0;
  b.tiny_wgs = own_tiny_launch ? 0 : (b.n_tiny + kWaves * R * TinyT<L>::value - 1) / (kWaves * R * TinyT<L>::value);
  b.free_wgs_pp = b.wide_wgs + (b.n_tasks - b.n_tiny - b.n_wide + kWaves * R - 1) / (kWaves * R) + b.tiny_wgs;
  // the sliced region: per XCD ceil(slice_xcd_tasks / tasks per workgroup) workgroups, interleaved b = x (mod 8); a panel is
  // padded to a multiple of 8 workgroups so that b mod 8 == blockIdx mod 8 in every panel (the idle ones return at once)
  b.slice_wgs = a.n_slices > 0 ? 8 * ((a.slice_xcd_tasks + kWaves * R - 1) / (kWaves * R)) : 0;
  b.sparse_wgs_pp = b.slice_wgs + b.free_wgs_pp;
  if (b.slice_wgs > 0) b.sparse_wgs_pp = (b.sparse_wgs_pp + 7) & ~7;
  const int n_col_panels = (a.D + a.panel_cols - 1) / a.panel_cols;
  b.sparse_wgs = b.sparse_wgs_pp * n_col_panels;
  if (b.sparse_wgs_pp == 0) b.sparse_wgs_pp = 1;  // divisor in the kernel
  // dense-tile panel width: 16*dense_vec columns -- the narrowest of the three lane widths that covers the embedding in ONE panel
  // (a second panel walks the window's column list and gathers its rows again: D = 48 as 32 + 16 columns took 382 us on the
  // YeastH-sized graph where D = 50 in one 64-column panel takes 279, profiles/r04/ab_dense_panels.log), else the widest.  A 16-bit
  // build wider than one element per lane is only launched on even widths (capi.hip pick_vec), so its narrower lanes, the last one
  // moved back, stay on the dword grid.
  constexpr int VM = DenseV<VEC>::mid;
  b.dense_vec = a.D <= 16 ? 1 : (a.D <= 16 * VM ? VM : VEC);
  b.n_panels = (a.D + 16 * b.dense_vec - 1) / (16 * b.dense_vec);
  constexpr bool kCanFuse = sizeof(typename E::T) == 4 && VEC == 4;
  constexpr int kMinWaves = sizeof(typename E::T) == 4 ? HCSPMM_MIN_WAVES_PER_SIMD : HCSPMM_MIN_WAVES_H16;
  if (dense_fused && !kCanFuse) return hipErrorInvalidValue;  // the caller checked (capi.hip fused_single_launch_ok)
  const long long dense_units = (long long)b.n_dense * (dense_fused ? 1 : b.n_panels);  // fused: a wave owns a window
  long long dense_wgs = (dense_units + kWaves - 1) / kWaves;
  if (dense_fused) {
    static const long long cap = [] {
      const char* e = getenv("HCSPMM_FUSED_DENSE_WGS");
      const long long v = e ? atoll(e) : 0;
      return v > 0 ? v : 2048LL;
    }();
    if (dense_wgs > cap) dense_wgs = cap;
  }
  b.fused_dense_wgs = (int)dense_wgs;
  const long long grid = (long long)b.sparse_wgs + dense_wgs;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (grid > 0) {
    if constexpr (kCanFuse) {
      if (dense_fused) {
        const size_t lds = (size_t)b.n_panels * 16 * b.dense_vec * fused_row_stride(a.H, b.dense_vec) * sizeof(float);
        // (four waves per SIMD, like the plain kernel: at three -- 137 registers, nothing spilled -- the in-launch form LOSES 2-10 %: profiles/r02/ab_fused.log)
        hipLaunchKernelGGL((hybrid_plan_kernel<E, L, VEC, HCSPMM_SPARSE_U, HCSPMM_FUSED_MIN_WAVES, true>),
                           dim3((unsigned)grid), dim3(kThreads), lds, stream, b);
      } else {
        hipLaunchKernelGGL((hybrid_plan_kernel<E, L, VEC, HCSPMM_SPARSE_U, kMinWaves>), dim3((unsigned)grid),
                           dim3(kThreads), 0, stream, b);
      }
    } else {
      hipLaunchKernelGGL((hybrid_plan_kernel<E, L, VEC, HCSPMM_SPARSE_U, kMinWaves>), dim3((unsigned)grid),
                         dim3(kThreads), 0, stream, b);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (b.tiny_kernel_wgs > 0) {
    hipLaunchKernelGGL((tiny_kernel<E, L, VEC>), dim3((unsigned)(b.tiny_kernel_wgs * n_col_panels)), dim3(kThreads), 0, stream, b);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (a.n_split_rows > 0) {
    const int fg = (a.n_split_rows + kWaves - 1) / kWaves;
    hipLaunchKernelGGL((fixup_kernel<E, VEC>), dim3(fg), dim3(kThreads), 0, stream, b);
    e = hipGetLastError();
  }
  return e;
}

template <typename E, int L, int VEC>
static hipError_t launch_window_LV(const WindowArgs& a, hipStream_t stream) {
  const int W = (a.N + 15) / 16;
  // 16 rows x L lanes of sparse work, D/(16*VEC) dense panels: narrow embeddings get narrower workgroups
  // (a 256-thread workgroup per window is dispatch-bound when each window holds a handful of entries)
  const int n_panels = (a.D + 16 * VEC - 1) / (16 * VEC);
  int waves = (16 * L + 63) / 64;
  if (n_panels > waves) waves = n_panels;
  if (waves > kWaves) waves = kWaves;
  if (W > 0) hipLaunchKernelGGL((hybrid_window_kernel<E, L, VEC>), dim3(W), dim3(waves * 64), 0, stream, a);
  return hipGetLastError();
}

// lanes per task: smallest power of two >= D / VEC, clamped to [4, 64]
static inline int pick_L(int D, int VEC) {
  const int slots = (D + VEC - 1) / VEC;
  int L = 4;
  while (L < slots && L < 64) L <<= 1;
  return L;
}

#define HCSPMM_DISPATCH_L(FN, E, VEC, WIDTH, ARGS, STREAM)  \
  switch (pick_L((WIDTH), VEC)) {                           \
    case 4:  return FN<E, 4, VEC>(ARGS, STREAM);            \
    case 8:  return FN<E, 8, VEC>(ARGS, STREAM);            \
    case 16: return FN<E, 16, VEC>(ARGS, STREAM);           \
    case 32: return FN<E, 32, VEC>(ARGS, STREAM);           \
    default: return FN<E, 64, VEC>(ARGS, STREAM);           \
  }

}  // namespace hcspmm
