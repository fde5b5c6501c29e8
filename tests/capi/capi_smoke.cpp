// capi_smoke.cpp -- a consumer of libhcspmm.so with NO Python and NO torch: plain C++ + the HIP
// runtime, exactly what include/hcspmm.h promises (raw pointers, sizes, a stream).  Builds a small
// graph, runs the host preprocess + plan, uploads, calls hcspmm_forward on its own stream (planned and
// plan-free), and checks both against a host loop with integer-valued features (exact in fp32).
// Built and run by tests/test_capi_native.py:  hipcc capi_smoke.cpp -I include -L csrc -lhcspmm
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "hcspmm.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define HC_OK(x)                                                                              \
  do {                                                                                        \
    int r_ = (x);                                                                             \
    if (r_ != HCSPMM_OK) {                                                                    \
      std::fprintf(stderr, "hcspmm error %d (%s) at %s:%d\n", r_, hcspmm_strerror(r_), __FILE__, __LINE__); \
      return 3;                                                                               \
    }                                                                                         \
  } while (0)

template <typename T>
static T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  const int64_t N = 1000 + 7;  // not a multiple of 16
  const int D = 48;
  // graph: planted 16-row groups sharing 12 columns (dense-tile windows), every 5th window random and wider
  // (sparse-row windows), one hub row (split + fix-up)
  std::vector<int32_t> rowptr(N + 1, 0), col;
  uint64_t rng = 88172645463325252ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  for (int64_t w = 0; w * 16 < N; ++w) {
    std::vector<int32_t> shared_cols;
    for (int k = 0; k < 12; ++k) shared_cols.push_back((int32_t)(next() % N));
    for (int64_t r = w * 16; r < N && r < w * 16 + 16; ++r) {
      std::vector<char> seen(N, 0);
      std::vector<int32_t> mine;
      auto add = [&](int32_t c) { if (!seen[c]) { seen[c] = 1; mine.push_back(c); } };
      if (r == 3) for (int k = 0; k < 900; ++k) add((int32_t)(next() % N));          // hub row
      else if (w % 5 == 4) for (int k = 0; k < 9; ++k) add((int32_t)(next() % N));   // unstructured window
      else for (int32_t c : shared_cols) if (next() % 2) add(c);
      std::vector<int32_t> sorted;
      for (int64_t c = 0; c < N; ++c) if (seen[c]) sorted.push_back((int32_t)c);
      col.insert(col.end(), sorted.begin(), sorted.end());
      rowptr[r + 1] = (int32_t)col.size();
    }
  }
  const int64_t E = (int64_t)col.size(), W = (N + 15) / 16;
  std::vector<int32_t> bp(W), ht(W), e2c(E), e2r(E);
  HC_OK(hcspmm_preprocess_host(rowptr.data(), col.data(), N, E, N, HCSPMM_RULE_INTENDED, 2, bp.data(), e2c.data(), e2r.data(),
                               ht.data()));
  int64_t words = 0;
  hcspmm_plan_params pp = {256, 128, 0};  // make the hub row split
  HC_OK(hcspmm_plan_words(rowptr.data(), N, E, bp.data(), ht.data(), &pp, &words));
  std::vector<int32_t> plan((size_t)words);
  HC_OK(hcspmm_plan_build(rowptr.data(), col.data(), N, E, N, bp.data(), e2c.data(), ht.data(), &pp, plan.data(), words));
  hcspmm_plan_header header;
  std::memcpy(&header, plan.data(), sizeof(header));
  HC_OK(hcspmm_plan_check(&header, N, E, words));
  if (header.n_dense == 0 || header.n_tasks == 0 || header.n_split_rows != 1) {
    std::fprintf(stderr, "unexpected plan: dense %d tasks %d split %d\n", header.n_dense, header.n_tasks, header.n_split_rows);
    return 4;
  }

  std::vector<float> X((size_t)N * D), want((size_t)N * D, 0.0f);
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < D; ++d) X[(size_t)i * D + d] = (float)((i * 7 + d) % 61);
  for (int64_t r = 0; r < N; ++r)
    for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e)
      for (int d = 0; d < D; ++d) want[(size_t)r * D + d] += X[(size_t)col[e] * D + d];

  int32_t *rp_d = upload(rowptr), *col_d = upload(col), *bp_d = upload(bp), *ht_d = upload(ht), *e2c_d = upload(e2c),
          *e2r_d = upload(e2r), *plan_d = upload(plan);
  float* X_d = upload(X);
  float* Z_d = nullptr;
  HIP_OK(hipMalloc(&Z_d, sizeof(float) * (size_t)N * D));
  const size_t ws_bytes = hcspmm_workspace_bytes(&header, D);
  void* ws_d = nullptr;
  if (ws_bytes) HIP_OK(hipMalloc(&ws_d, ws_bytes));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  std::vector<float> got((size_t)N * D);
  for (int pass = 0; pass < 2; ++pass) {  // 0: planned, 1: plan-free (the reference's placeholder convention)
    HIP_OK(hipMemsetAsync(Z_d, 0xff, sizeof(float) * (size_t)N * D, stream));
    HC_OK(hcspmm_forward(X_d, Z_d, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, pass == 0 ? plan_d : nullptr,
                         pass == 0 ? &header : nullptr, N, E, D, pass == 0 ? ws_d : nullptr, pass == 0 ? ws_bytes : 0,
                         (void*)stream));
    HIP_OK(hipMemcpyAsync(got.data(), Z_d, sizeof(float) * got.size(), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < got.size(); ++i)
      if (got[i] != want[i]) {
        std::fprintf(stderr, "pass %d mismatch at %zu: %g vs %g\n", pass, i, got[i], want[i]);
        return 5;
      }
  }
  // fused aggregate + update (hcspmm_forward_fused: out2 = A*X, out = out2*W) in the two-launch form and in the row-tile
  // form (a plan built with fuse_in_launch = 2): small integer operands, so both must be EXACT and identical
  {
    const int H = 32;
    std::vector<float> Wm((size_t)D * H), want_out((size_t)N * H, 0.0f);
    for (int d = 0; d < D; ++d)
      for (int h = 0; h < H; ++h) Wm[(size_t)d * H + h] = (float)((d + 2 * h) % 5 - 2);
    for (int64_t r = 0; r < N; ++r)
      for (int h = 0; h < H; ++h) {
        double acc = 0.0;
        for (int d = 0; d < D; ++d) acc += (double)want[(size_t)r * D + d] * (double)Wm[(size_t)d * H + h];
        want_out[(size_t)r * H + h] = (float)acc;  // |acc| < 2^24: exact
      }
    float* W_d = upload(Wm);
    float* out_d = nullptr;
    HIP_OK(hipMalloc(&out_d, sizeof(float) * (size_t)N * H));
    std::vector<float> got_out((size_t)N * H);
    for (int form = 0; form <= 2; form += 2) {
      hcspmm_plan_params pf = {256, 128, form == 0 ? -1 : 2};
      int64_t fw = 0;
      HC_OK(hcspmm_plan_words(rowptr.data(), N, E, bp.data(), ht.data(), &pf, &fw));
      std::vector<int32_t> fplan((size_t)fw);
      HC_OK(hcspmm_plan_build(rowptr.data(), col.data(), N, E, N, bp.data(), e2c.data(), ht.data(), &pf, fplan.data(), fw));
      hcspmm_plan_header fh;
      std::memcpy(&fh, fplan.data(), sizeof(fh));
      if (hcspmm_fused_in_launch(&fh, D, H) != form) {
        std::fprintf(stderr, "fused form %d expected, got %d\n", form, hcspmm_fused_in_launch(&fh, D, H));
        return 14;
      }
      int32_t* fplan_d = upload(fplan);
      HIP_OK(hipMemsetAsync(Z_d, 0xff, sizeof(float) * (size_t)N * D, stream));
      HIP_OK(hipMemsetAsync(out_d, 0xff, sizeof(float) * (size_t)N * H, stream));
      HC_OK(hcspmm_forward_fused(X_d, out_d, Z_d, W_d, H, 1, H, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, fplan_d, &fh, N, E, D, ws_d,
                                 ws_bytes, (void*)stream));
      HIP_OK(hipMemcpyAsync(got.data(), Z_d, sizeof(float) * got.size(), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemcpyAsync(got_out.data(), out_d, sizeof(float) * got_out.size(), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      for (size_t i = 0; i < got.size(); ++i)
        if (got[i] != want[i]) {
          std::fprintf(stderr, "fused form %d: out2 mismatch at %zu: %g vs %g\n", form, i, got[i], want[i]);
          return 15;
        }
      for (size_t i = 0; i < got_out.size(); ++i)
        if (got_out[i] != want_out[i]) {
          std::fprintf(stderr, "fused form %d: out mismatch at %zu: %g vs %g\n", form, i, got_out[i], want_out[i]);
          return 16;
        }
      HIP_OK(hipFree(fplan_d));
    }
    HIP_OK(hipFree(W_d));
    HIP_OK(hipFree(out_d));
  }
  // bfloat16 features through hcspmm_forward_typed: 16-bit X and Z, fp32 accumulation, one rounding (to nearest
  // even) per output element.  The features are small integers (exact in bf16), so the fp32 sum is exact and
  // the expected output is simply the rounded `want`.
  {
    auto to_bf16 = [](float f) {
      uint32_t u;
      std::memcpy(&u, &f, 4);
      u += 0x7fffu + ((u >> 16) & 1u);
      return (uint16_t)(u >> 16);
    };
    std::vector<uint16_t> X16(X.size()), got16(X.size());
    for (size_t i = 0; i < X.size(); ++i) X16[i] = to_bf16(X[i]);
    uint16_t* X16_d = upload(X16);
    uint16_t* Z16_d = nullptr;
    HIP_OK(hipMalloc(&Z16_d, sizeof(uint16_t) * X.size()));
    for (int pass = 0; pass < 2; ++pass) {
      HIP_OK(hipMemsetAsync(Z16_d, 0xff, sizeof(uint16_t) * X.size(), stream));
      HC_OK(hcspmm_forward_typed(X16_d, N, D, Z16_d, D, HCSPMM_DTYPE_BF16, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d,
                                 pass == 0 ? plan_d : nullptr, pass == 0 ? &header : nullptr, N, E, D,
                                 pass == 0 ? ws_d : nullptr, pass == 0 ? ws_bytes : 0, (void*)stream));
      HIP_OK(hipMemcpyAsync(got16.data(), Z16_d, sizeof(uint16_t) * got16.size(), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      for (size_t i = 0; i < got16.size(); ++i)
        if (got16[i] != to_bf16(want[i])) {
          std::fprintf(stderr, "bf16 pass %d mismatch at %zu: %04x vs %04x\n", pass, i, got16[i], to_bf16(want[i]));
          return 8;
        }
    }
    if (hcspmm_forward_typed(X16_d, N, D, Z16_d, D, 7, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, plan_d, &header, N, E, D, ws_d,
                             ws_bytes, stream) != HCSPMM_EINVAL)
      return 9;
  }
  // argument errors come back as codes, not crashes
  if (hcspmm_forward(X_d, Z_d, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, plan_d, &header, N + 1, E, D, ws_d, ws_bytes, stream) !=
      HCSPMM_EPLAN)
    return 6;
  if (ws_bytes && hcspmm_forward(X_d, Z_d, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, plan_d, &header, N, E, D, nullptr, 0, stream) !=
                      HCSPMM_EWORKSPACE)
    return 7;
  // a column id outside [0, num_columns) is refused on the host (it would be an out-of-bounds gather on the GPU)
  {
    std::vector<int32_t> bad_col = col;
    bad_col[E / 2] = (int32_t)N;  // one past the last row of X
    std::vector<int32_t> bp2(W), ht2(W), e2c2(E);
    if (hcspmm_preprocess_host(rowptr.data(), bad_col.data(), N, E, N, HCSPMM_RULE_INTENDED, 2, bp2.data(), e2c2.data(), nullptr,
                               ht2.data()) != HCSPMM_EINVAL)
      return 10;
    std::vector<int32_t> plan2((size_t)words);
    if (hcspmm_plan_build(rowptr.data(), bad_col.data(), N, E, N, bp.data(), e2c.data(), ht.data(), &pp, plan2.data(), words) !=
        HCSPMM_EINVAL)
      return 11;
    // ... while the same id is fine for a row block whose columns index a taller X (num_columns = 2N)
    if (hcspmm_plan_build(rowptr.data(), bad_col.data(), N, E, 2 * N, bp.data(), e2c.data(), ht.data(), &pp, plan2.data(), words) !=
        HCSPMM_OK)
      return 12;
    hcspmm_plan_header h2;
    std::memcpy(&h2, plan2.data(), sizeof(h2));
    // and a launch with that plan over an X of only N rows is refused before anything is enqueued
    if (h2.num_columns != 2 * N ||
        hcspmm_forward_strided(X_d, N, D, Z_d, D, rp_d, col_d, bp_d, e2c_d, e2r_d, ht_d, plan_d, &h2, N, E, D, ws_d, ws_bytes,
                               stream) != HCSPMM_EINVAL)
      return 13;
  }
  // edgeToRow made on the device (fill_edgeToRow's counterpart) equals the host pass's
  {
    int32_t* e2r2_d = nullptr;
    HIP_OK(hipMalloc(&e2r2_d, sizeof(int32_t) * (size_t)E));
    HC_OK(hcspmm_edge_to_row_device(rp_d, N, E, e2r2_d, (void*)stream));
    std::vector<int32_t> e2r2((size_t)E);
    HIP_OK(hipMemcpyAsync(e2r2.data(), e2r2_d, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (e2r2 != e2r) return 15;
  }
  // the device fingerprint of the uploaded graph equals the host one stored in the plan header
  {
    uint64_t host_fp = 0, dev_fp = 0, *fp_d = nullptr;
    HC_OK(hcspmm_graph_fingerprint_host(rowptr.data(), col.data(), N, E, &host_fp));
    HIP_OK(hipMalloc(&fp_d, sizeof(uint64_t)));
    HC_OK(hcspmm_graph_fingerprint_device(rp_d, col_d, N, E, fp_d, (void*)stream));
    HIP_OK(hipMemcpyAsync(&dev_fp, fp_d, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    const uint64_t in_header = ((uint64_t)header.fingerprint_hi << 32) | header.fingerprint_lo;
    if (host_fp != dev_fp || host_fp != in_header) {
      std::fprintf(stderr, "fingerprint mismatch: host %llx device %llx header %llx\n", (unsigned long long)host_fp,
                   (unsigned long long)dev_fp, (unsigned long long)in_header);
      return 14;
    }
  }
  std::printf("capi_smoke ok: N=%lld E=%lld dense_windows=%d tasks=%d split_rows=%d\n", (long long)N, (long long)E,
              header.n_dense, header.n_tasks, header.n_split_rows);
  return 0;
}
