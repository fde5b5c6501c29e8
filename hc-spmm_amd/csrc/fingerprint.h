// fingerprint.h -- the terms of hcspmm_graph_fingerprint_* (include/hcspmm.h), shared by the host pass
// (plan_host.cpp) and the device kernel (capi.hip) so that both add up exactly the same 64-bit numbers.
//   fingerprint = mix(N, E) + sum_r term_rowptr(r, rowptr[r]) + sum_e term_col(e, col[e])   (mod 2^64)
// A sum is order-independent, so every host thread / GPU wave adds its own share.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HCSPMM_HD __host__ __device__
#else
#define HCSPMM_HD
#endif

namespace hcspmm {

HCSPMM_HD inline uint64_t fp_mix(uint64_t x) {  // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
HCSPMM_HD inline uint64_t fp_term_col(uint64_t e, int32_t c) { return fp_mix((e << 32) | (uint32_t)c); }
HCSPMM_HD inline uint64_t fp_term_rowptr(uint64_t r, int32_t v) { return fp_mix(~((r << 32) | (uint32_t)v)); }
HCSPMM_HD inline uint64_t fp_seed(int64_t N, int64_t E) { return fp_mix(((uint64_t)N << 32) ^ (uint64_t)E ^ 0x4843535000000000ull); }

}  // namespace hcspmm
