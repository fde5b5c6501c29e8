// host_util.h -- shared by the host-side translation units (preprocess_host.cpp, plan_host.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>

namespace hcspmm {

// Worker threads for the host passes: HCSPMM_THREADS if set (>= 1), else min(64, hardware threads).
// Every pass assigns contiguous window ranges to threads and derives output positions from per-range
// counts, so results do not depend on this number.
inline int host_threads() {
  static const int n = [] {
    if (const char* e = std::getenv("HCSPMM_THREADS")) {
      const int v = std::atoi(e);
      if (v >= 1) return std::min(v, 256);
    }
    const int hw = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(64, hw));
  }();
  return n;
}

// CSR row pointers as every host entry point needs them before it reads a single column id: start at 0, end at E,
// never decrease (so that every [rowptr[r], rowptr[r+1]) lies inside column_index[0, E)).
inline bool csr_row_pointers_ok(const int32_t* rowptr, int64_t N, int64_t E) {
  if (rowptr[0] != 0 || rowptr[N] != E) return false;
  for (int64_t r = 0; r < N; ++r)
    if (rowptr[r + 1] < rowptr[r]) return false;
  return true;
}

}  // namespace hcspmm
