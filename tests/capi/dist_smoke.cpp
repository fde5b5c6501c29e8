// dist_smoke.cpp -- a consumer of libhcspmm_dist.so (include/hcspmm_dist.h) with NO Python and NO torch: plain C++ + HIP +
// RCCL.  One process per GPU; on a one-GPU box the communicator has ONE rank (ncclCommInitAll over device 0) and the
// collective path is forced (always_gather), which exercises everything but the wire: RCCL initialisation, ncclAllGather
// on the communication stream, the event ordering between the two streams over several steps with changing features, the
// panel-major buffers with padding rows, and the strided products.  Integer-valued features: results are exact.
// With more GPUs visible (argv[1] = world) it forks one process per GPU and runs the real exchange (tests/test_capi_native.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hcspmm_dist.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "RCCL error %d at %s:%d\n", (int)r_, __FILE__, __LINE__); return 4; } } while (0)
#define HC_OK(x) do { int r_ = (x); if (r_ != HCSPMM_OK) { std::fprintf(stderr, "hcspmm error %d (%s, detail %d) at %s:%d\n", r_, hcspmm_strerror(r_), hcspmm_dist_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <typename T> static T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())) != hipSuccess) return nullptr;
  if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

// the whole graph (every rank builds the same one): N rows, a few hubs, unsorted draws made ascending + unique per row
static void make_graph(int64_t N, std::vector<int32_t>& rowptr, std::vector<int32_t>& col) {
  uint64_t rng = 88172645463325252ull;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  rowptr.assign((size_t)N + 1, 0);
  std::vector<char> seen((size_t)N);
  for (int64_t r = 0; r < N; ++r) {
    const int deg = (r % 977 == 5) ? 700 : (int)(next() % 12);
    std::fill(seen.begin(), seen.end(), 0);
    for (int k = 0; k < deg; ++k) seen[(size_t)(next() % (uint64_t)N)] = 1;
    for (int64_t c = 0; c < N; ++c) if (seen[(size_t)c]) col.push_back((int32_t)c);
    rowptr[(size_t)r + 1] = (int32_t)col.size();
  }
}

// the communicator's unique id travels from rank 0 to the others through a file (written whole, then renamed)
static int exchange_id(int rank, const char* path, ncclUniqueId* id) {
  if (rank == 0) {
    NCCL_OK(ncclGetUniqueId(id));
    char tmp[512];
    std::snprintf(tmp, sizeof(tmp), "%s.tmp", path);
    FILE* f = std::fopen(tmp, "wb");
    if (!f || std::fwrite(id, sizeof(*id), 1, f) != 1) return 8;
    std::fclose(f);
    if (std::rename(tmp, path) != 0) return 8;
    return 0;
  }
  for (int tries = 0; tries < 3000; ++tries) {  // up to 5 minutes
    FILE* f = std::fopen(path, "rb");
    if (f) {
      const size_t n = std::fread(id, sizeof(*id), 1, f);
      std::fclose(f);
      if (n == 1) return 0;
    }
    usleep(100000);
  }
  return 8;
}

static int run_rank(int rank, int world, const char* id_path) {
  HIP_OK(hipSetDevice(rank));
  ncclComm_t comm;
  if (world == 1) { int dev0 = 0; NCCL_OK(ncclCommInitAll(&comm, 1, &dev0)); }
  else {
    ncclUniqueId id;
    const int rc = exchange_id(rank, id_path, &id);
    if (rc) return rc;
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));
  }
  const int64_t N = 6000 + 11, D = 64;
  const int n_panels = 4, w = (int)D / n_panels;
  std::vector<int32_t> rowptr, col;
  make_graph(N, rowptr, col);
  // the library's own partition (nnz-balanced, window-aligned: blocks of unequal height, so every rank but the tallest
  // carries padding rows) and block extraction (column ids remapped to rows of the padded gathered matrix)
  std::vector<int64_t> ranges((size_t)2 * world);
  HC_OK(hcspmm_dist_partition_rows(rowptr.data(), N, world, ranges.data()));
  const int64_t r0 = ranges[(size_t)2 * rank], r1 = ranges[(size_t)2 * rank + 1], n_local = r1 - r0;
  std::vector<int32_t> rp_l((size_t)n_local + 1), col_l((size_t)(rowptr[(size_t)r1] - rowptr[(size_t)r0]));
  int64_t pad = 0;
  HC_OK(hcspmm_dist_extract_block(rowptr.data(), col.data(), N, world, ranges.data(), rank, rp_l.data(), col_l.data(), &pad));
  const int64_t M = (int64_t)world * pad;  // rows of the gathered matrix: global vertex v sits at owner(v) * pad + (v - first row of owner(v))
  const int64_t E = (int64_t)col_l.size(), W = (n_local + 15) / 16;
  std::vector<int32_t> bp((size_t)W), ht((size_t)W), e2c((size_t)E), e2r((size_t)E);
  HC_OK(hcspmm_preprocess_host(rp_l.data(), col_l.data(), n_local, E, M, HCSPMM_RULE_INTENDED, 2, bp.data(), e2c.data(), e2r.data(), ht.data()));
  int64_t words = 0;
  hcspmm_plan_params pp = {256, 128, 0, 64, 8};  // hub rows: XCD-affine column slices + segments
  HC_OK(hcspmm_plan_words(rp_l.data(), n_local, E, bp.data(), ht.data(), &pp, &words));
  std::vector<int32_t> plan((size_t)words);
  HC_OK(hcspmm_plan_build(rp_l.data(), col_l.data(), n_local, E, M, bp.data(), e2c.data(), ht.data(), &pp, plan.data(), words));
  hcspmm_plan_header header;
  std::memcpy(&header, plan.data(), sizeof(header));
  plan.resize((size_t)header.total_words);
  int32_t *rp_d = upload(rp_l), *col_d = upload(col_l), *bp_d = upload(bp), *ht_d = upload(ht), *e2c_d = upload(e2c), *e2r_d = upload(e2r),
          *plan_d = upload(plan);
  float *x_pm = nullptr, *g_pm = nullptr, *z_pm = nullptr;
  HIP_OK(hipMalloc(&x_pm, sizeof(float) * (size_t)n_panels * pad * w));
  HIP_OK(hipMalloc(&g_pm, sizeof(float) * (size_t)n_panels * M * w));
  HIP_OK(hipMalloc(&z_pm, sizeof(float) * (size_t)n_panels * std::max<int64_t>(n_local, 1) * w));
  const size_t ws_bytes = hcspmm_workspace_bytes(&header, w);
  void* ws_d = nullptr;
  if (ws_bytes) HIP_OK(hipMalloc(&ws_d, ws_bytes));
  hipStream_t compute, comm_stream;
  HIP_OK(hipStreamCreate(&compute));
  HIP_OK(hipStreamCreate(&comm_stream));
  hcspmm_dist_ctx* ctx = nullptr;
  HC_OK(hcspmm_dist_create(n_panels, &ctx));
  hcspmm_dist_step_args a;
  std::memset(&a, 0, sizeof(a));
  a.nccl_comm = comm; a.world_size = world; a.n_panels = n_panels; a.embedding_dim = (int)D; a.dtype = HCSPMM_DTYPE_F32;
  a.pad_rows = pad; a.n_local = n_local; a.num_edges = E; a.x_pm = x_pm; a.gathered_pm = g_pm; a.z_pm = z_pm;
  a.row_pointers_d = rp_d; a.column_index_d = col_d; a.blockPartition_d = bp_d; a.edgeToColumn_d = e2c_d; a.edgeToRow_d = e2r_d;
  a.hybrid_type_d = ht_d; a.plan_d = plan_d; a.plan_header_h = &header; a.workspace_d = ws_d; a.workspace_bytes = ws_bytes;
  a.compute_stream = compute; a.comm_stream = comm_stream; a.always_gather = 1;
  std::vector<float> xh((size_t)n_panels * pad * w), got((size_t)n_panels * std::max<int64_t>(n_local, 1) * w);
  // features of global vertex v at step s: X[v][d] = (v * 7 + d * 3 + s * 5) % 61 -- every step differs, so a product that ran
  // ahead of its gather (or a gather that overtook the previous product) shows up as a wrong sum
  auto feat = [](int64_t v, int64_t d, int s) { return (float)((v * 7 + d * 3 + s * 5) % 61); };
  for (int s = 0; s < 4; ++s) {
    std::fill(xh.begin(), xh.end(), 0.0f);
    for (int p = 0; p < n_panels; ++p)
      for (int64_t r = 0; r < n_local; ++r)
        for (int c = 0; c < w; ++c) xh[((size_t)p * pad + r) * w + c] = feat(r0 + r, p * w + c, s);
    HIP_OK(hipMemcpyAsync(x_pm, xh.data(), sizeof(float) * xh.size(), hipMemcpyHostToDevice, compute));  // a producer on the compute stream
    HC_OK(hcspmm_dist_step(ctx, &a));
    if (s == 1) HC_OK(hcspmm_dist_step(ctx, &a));  // back-to-back steps on the same buffers
    HIP_OK(hipMemcpyAsync(got.data(), z_pm, sizeof(float) * got.size(), hipMemcpyDeviceToHost, compute));
    HIP_OK(hipStreamSynchronize(compute));
    for (int64_t r = 0; r < n_local; ++r)
      for (int64_t d = 0; d < D; ++d) {
        float want = 0.0f;
        for (int32_t e = rowptr[(size_t)(r0 + r)]; e < rowptr[(size_t)(r0 + r) + 1]; ++e) want += feat(col[(size_t)e], d, s);
        const float g = got[((size_t)(d / w) * n_local + r) * w + d % w];
        if (g != want) { std::fprintf(stderr, "rank %d step %d row %lld col %lld: %g vs %g\n", rank, s, (long long)r, (long long)d, g, want); return 5; }
      }
  }
  if (world == 1) {  // without the forced collective a one-rank shard multiplies x_pm in place: same result
    a.always_gather = 0;
    HC_OK(hcspmm_dist_step(ctx, &a));
    std::vector<float> got2(got.size());
    HIP_OK(hipMemcpyAsync(got2.data(), z_pm, sizeof(float) * got2.size(), hipMemcpyDeviceToHost, compute));
    HIP_OK(hipStreamSynchronize(compute));
    if (got2 != got) return 6;
  }
  a.n_panels = 3;  // 64 % 3 != 0
  if (hcspmm_dist_step(ctx, &a) != HCSPMM_EINVAL) return 7;
  hcspmm_dist_destroy(ctx);
  ncclCommDestroy(comm);
  std::printf("dist_smoke ok: rank %d of %d, rows [%lld, %lld), %lld entries, %d column slices, %d split rows\n", rank, world,
              (long long)r0, (long long)r1, (long long)E, header.n_slices, header.n_split_rows);
  return 0;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? std::atoi(argv[1]) : 1;
  if (world <= 1) return run_rank(0, 1, nullptr);
  // one process per GPU, forked BEFORE anything here has touched a GPU or RCCL
  char id_path[256];
  std::snprintf(id_path, sizeof(id_path), "/tmp/hcspmm_dist_smoke_%d.id", (int)getpid());
  std::remove(id_path);
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {
    const pid_t pid = fork();
    if (pid == 0) _exit(run_rank(r, world, id_path));
    kids.push_back(pid);
  }
  int rc = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : (WIFEXITED(st) ? WEXITSTATUS(st) : 99);
  }
  std::remove(id_path);
  return rc;
}
