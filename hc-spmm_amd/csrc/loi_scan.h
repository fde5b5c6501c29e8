// loi_scan.h -- the pricing scan of the LOI reorder (LOI.cpp:775-781), shared by the exact (loi_host.cpp) and the
// relaxed parallel variant (loi_fast_host.cpp): same IEEE conversions and division in both.
#pragma once
#include <cstdint>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace hcspmm {
namespace loi {

// Position of the first candidate with the largest profit (float)(ones + deg) / (float)(base + deg - shared),
// strictly greater than every earlier one (and than 0); -1 if none is alive.  LOI.cpp:775-781.
inline int64_t scan_best_scalar(const int32_t* deg, const int32_t* shared, const int32_t* alive, int64_t n, int32_t ones,
                         int32_t base) {
  int64_t best = -1;
  float best_profit = 0.0f;
  for (int64_t i = 0; i < n; ++i) {
    if (!alive[i]) continue;
    const float profit = (float)(ones + deg[i]) / (float)(base + deg[i] - shared[i]);
    if (profit > best_profit) {
      best = i;
      best_profit = profit;
    }
  }
  return best;
}

inline bool have_avx2() {
#if defined(__x86_64__)
  return __builtin_cpu_supports("avx2");
#else
  return false;
#endif
}

#if defined(__x86_64__)
// Eight candidates per step: the same int -> float conversions (round to nearest even) and the same IEEE
// division as the scalar loop; a block is walked lane by lane only when one of its profits beats the running
// best, so the winner is still the FIRST position holding the maximum.
__attribute__((target("avx2"))) inline int64_t scan_best_avx2(const int32_t* deg, const int32_t* shared, const int32_t* alive,
                                                        int64_t n, int32_t ones, int32_t base) {
  int64_t best = -1;
  float best_profit = 0.0f;
  const __m256i v_ones = _mm256_set1_epi32(ones), v_base = _mm256_set1_epi32(base);
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(deg + i));
    const __m256i sh = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(shared + i));
    const __m256 al = _mm256_castsi256_ps(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(alive + i)));
    const __m256 num = _mm256_cvtepi32_ps(_mm256_add_epi32(v_ones, d));
    const __m256 den = _mm256_cvtepi32_ps(_mm256_sub_epi32(_mm256_add_epi32(v_base, d), sh));
    const __m256 profit = _mm256_and_ps(_mm256_div_ps(num, den), al);  // dead lanes price at 0: never > best
    if (_mm256_movemask_ps(_mm256_cmp_ps(profit, _mm256_set1_ps(best_profit), _CMP_GT_OQ)) != 0) {
      alignas(32) float p[8];
      _mm256_store_ps(p, profit);
      for (int k = 0; k < 8; ++k)
        if (p[k] > best_profit) {
          best = i + k;
          best_profit = p[k];
        }
    }
  }
  for (; i < n; ++i) {
    if (!alive[i]) continue;
    const float profit = (float)(ones + deg[i]) / (float)(base + deg[i] - shared[i]);
    if (profit > best_profit) {
      best = i;
      best_profit = profit;
    }
  }
  return best;
}
#else
inline int64_t scan_best_avx2(const int32_t* deg, const int32_t* shared, const int32_t* alive, int64_t n, int32_t ones,
                       int32_t base) {
  return scan_best_scalar(deg, shared, alive, n, ones, base);
}
#endif

inline int64_t scan_best(const int32_t* deg, const int32_t* shared, const int32_t* alive, int64_t n, int32_t ones, int32_t base) {
  static const bool avx2 = have_avx2();
  return avx2 ? scan_best_avx2(deg, shared, alive, n, ones, base) : scan_best_scalar(deg, shared, alive, n, ones, base);
}

}  // namespace loi
}  // namespace hcspmm
