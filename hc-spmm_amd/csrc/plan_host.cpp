// plan_host.cpp -- host-side launch plan for the gfx950 hybrid kernel (hcspmm_plan_*).
//
// New design (nothing comparable in the reference, which launches one block per window and
// lets hub rows serialise, hybrid_all_kernel.cu:435-438): the classified windows are turned into
//   tasks   : one per sparse-path row (rows longer than split_threshold are cut into segments of
//             segment_len entries whose partial sums a fix-up pass adds in a fixed order); tasks are
//             sorted by descending power-of-two length class (row order inside a class) so that
//             (a) the 64/L tasks sharing a wave have trip counts within 2x, (b) the hardware
//             dispatcher sees the heaviest work first, (c) neighbouring tasks touch neighbouring memory;
//   dense   : per dense-path window, its ascending unique columns (K = 8*blockPartition, padded
//             with -1) and, per 4-column k-step, the 64-bit lane mask of the 16x4 0/1 tile in
//             v_mfma_f32_16x16x4_f32 A-operand order (lane = 16*(k%4) + row);
//   compact : dense windows of at most HCSPMM_COMPACT_K columns (the usual case on low-degree graphs) get a
//             fixed 64-word record instead -- [window, K/4, U[40], 10 x (mask lo, mask hi), pad] -- whose
//             address follows from the unit number, so a wave fetches everything it needs to start
//             gathering with ONE coalesced 256-byte load (dense_index + U + masks are two dependent loads);
//             windows of 48..HCSPMM_COMPACT2_K columns get the same in 128 words (two words per lane);
//   fixups  : (row, first partial slot, segment count) for every split row.
// The last n_tiny tasks (those of at most two entries: on low-degree graphs the great majority) carry
// their column indices INSIDE the descriptor -- (row or -(slot+1), index0, length, index1), absent
// indices -1 -- so that the kernel needs one memory round trip, not two, before it can gather.
//   sparse windows : ids of the windows that are NOT dense (ascending) -- the 16-row tiles that the update pass
//             of the fused operators still has to multiply by the weights (dense windows do it in the launch).
//   slices  : XCD-affine column slices (hcspmm.h n_slices).  The launch is bound by the gathered X rows that miss the
//             per-XCD L2, and workgroups b and b + 8 share an XCD (MI355X_MICROARCH.md, workgroup dispatch).  Rows
//             longer than slice_threshold are therefore cut where their ascending column ids cross S - 1 boundaries
//             (chosen so that every slice holds about the same number of those entries) and every segment_len
//             entries; the pieces of slice s go to their own task list, which the kernel serves with workgroups
//             b = s (mod 8) only.  The L2 of that XCD then sees 1/8 of the columns from those tasks -- on the
//             Reddit-scale headline a 3.7 MB share of each 30 MB panel of X -- instead of all of them
//             (tools/l2_hit_simulation.py: 56 % -> 72 % hits at the default threshold of 256 entries; measured 51 % ->
//             67 %, the launch 17 % faster and no longer bound by the fabric).  A column slice of a column-sorted row
//             is a contiguous CSR range, so the pieces are ordinary (row, first, length, slot) tasks whose partial
//             sums the fix-up pass adds in slice order = CSR order.
// Blob layout (int32 words): header[64] | tasks[n_tasks][4] | dense_index[n_dense][4] |
// dense_pack[...] | compact2[n_dense_compact2][128] | compact[n_dense_compact][64] | fixups[n_split_rows][4] |
// sparse_windows[n_sparse_windows] | slice_table[n_slices + 1] | slice_tasks[n_slice_tasks][4].
// All section offsets are multiples of 4 words, the compact sections' of 64.  With slices the sections are sized
// by upper bounds that do not depend on the column ids (hcspmm_plan_words has none): the header holds the real counts.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "fingerprint.h"
#include "hcspmm.h"
#include "host_util.h"

namespace {

constexpr int kSliceAuto = 0, kSliceOn = 1, kSliceOff = -1;
constexpr int32_t kSliceAutoThreshold = 256;      // rows longer than this are sliced: flat optimum 192-256 on every workload (profiles/r03/ab_slices.log)
constexpr int64_t kSliceAutoMinColumns = 65536;  // a 128-byte line per X row: below this a panel of X is within two L2s anyway
// ... and up to a quarter of a million columns (a panel of X within eight L2s) only launches with enough work gain: a
// 40-70 us launch loses 3-17 % to the extra region and the longer fix-up (profiles/r03/ab_slices_midsize.log)
constexpr int64_t kSliceAutoAlwaysColumns = 250000;
constexpr int64_t kSliceAutoMinEntries = 3000000;
constexpr int kSliceSample = 4;                  // the boundary histogram looks at every 4th entry of the long rows
constexpr int kSlicePad = 64;                    // slice lists are padded to whole waves of any lane-group count (4 waves x 16)

struct Resolved {
  int32_t split_threshold, segment_len, fuse_in_launch;
  int32_t slice_mode, slice_threshold, n_slices, panel_cols;
};

int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : dflt;
}

Resolved resolve(const hcspmm_plan_params* p) {
  Resolved r{512, 256, 0, kSliceAuto, kSliceAutoThreshold, 8, 0};
  if (p) {
    if (p->split_threshold > 0) r.split_threshold = p->split_threshold;
    if (p->segment_len > 0) r.segment_len = p->segment_len;
    r.fuse_in_launch = p->fuse_in_launch < 0 ? -1 : (p->fuse_in_launch > 2 ? 2 : p->fuse_in_launch);
  }
  if (r.segment_len > r.split_threshold) r.segment_len = r.split_threshold;
  int thr = p ? p->slice_threshold : 0, ns = p ? p->n_slices : 0;
  if (thr == 0) thr = env_int("HCSPMM_SLICE_THRESHOLD", 0);  // A/B runs; explicit parameters win
  if (ns == 0) ns = env_int("HCSPMM_SLICES", 0);
  if (thr > 0) { r.slice_mode = kSliceOn; r.slice_threshold = thr; }
  else if (thr < 0) r.slice_mode = kSliceOff;
  if (ns > 0) r.n_slices = std::min(64, (ns + 7) / 8 * 8);
  if (p && p->panel_cols != 0) r.panel_cols = p->panel_cols < 0 ? -1 : std::min(1 << 20, (p->panel_cols + 15) / 16 * 16);
  return r;
}

// pieces a sliced row of d entries can fall into, whatever its column ids: every boundary and every segment_len
// entries start at most one new piece
inline int64_t pieces_bound(int64_t d, const Resolved& rp) {
  return std::min<int64_t>(d, rp.n_slices - 1 + (d + rp.segment_len - 1) / rp.segment_len);
}

// Cut points of one sliced row: cuts[s] = offset (in [0, d]) of the first entry of slice s, cuts[S] = d; entries
// [e0, e0 + d) are cut where the (ascending) column ids cross bounds[0 .. S-2] (bounds[s] = first column id of slice s + 1).
// Unsorted columns still give a partition of the row (placement loses its meaning, the sum does not).
inline void slice_cuts(const int32_t* col, int32_t e0, int64_t d, const int32_t* bounds, int S, int32_t* cuts) {
  const int32_t* first = col + e0;
  int64_t lo = 0;
  cuts[0] = 0;
  for (int s = 0; s < S; ++s) {
    const int64_t hi = (s == S - 1) ? d : std::lower_bound(first + lo, first + d, bounds[s]) - first;
    cuts[s + 1] = (int32_t)hi;
    lo = hi;
  }
}

// The pieces of one sliced row from its cut points: every slice's range, cut every seg entries.
// fn(slice, first entry, length), in CSR order.
template <typename F>
inline void for_each_piece(const int32_t* cuts, int32_t e0, int S, int32_t seg, F fn) {
  for (int s = 0; s < S; ++s)
    for (int64_t b = cuts[s]; b < cuts[s + 1]; b += seg) fn(s, (int32_t)(e0 + b), (int32_t)std::min<int64_t>(seg, cuts[s + 1] - b));
}

inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

struct Layout {
  int64_t n_tasks = 0, n_dense = 0, n_compact = 0, n_compact2 = 0, n_split_rows = 0, n_partials = 0;
  int64_t dense_pack_words = 0;
  int64_t nnz_sparse = 0, nnz_dense = 0, dense_k_sum = 0;
  int32_t max_dense_k = 0;
  int64_t n_sparse_windows = 0;
  int64_t n_slice_tasks = 0, n_sliced_rows = 0, nnz_sliced = 0;  // with slices: n_slice_tasks, n_split_rows, n_partials are upper bounds
  int64_t off_tasks = 0, off_dense_index = 0, off_dense_pack = 0, off_compact2 = 0, off_compact = 0, off_fixups = 0,
          off_sparse_windows = 0, off_slice_table = 0, off_slice_tasks = 0, total = 0;
};

// The workers of ONE hcspmm_plan_build / hcspmm_plan_words call: started once, handed the call's passes one after the other (a
// pass used to start and join its own threads: eight start-ups of 16 threads were a quarter of the Reddit-scale plan build).
// Call-local -- created and joined inside the entry point, reached by the passes through a thread-local pointer -- so the library
// still keeps no state between calls.  Workers spin briefly, then sleep on a condition variable.
class PassPool {
 public:
  explicit PassPool(int T) : T_(T) {
    for (int t = 1; t < T_; ++t) th_.emplace_back([this, t] { work(t); });
  }
  ~PassPool() {
    fn_ = nullptr;
    post();
    for (auto& x : th_) x.join();
  }
  int size() const { return T_; }
  template <typename F> void run(F& fn) {
    fn_ = [](void* f, int t) { (*static_cast<F*>(f))(t); };
    arg_ = &fn;
    done_.store(0, std::memory_order_relaxed);
    post();
    fn(0);
    for (int spin = 0; done_.load(std::memory_order_acquire) != T_ - 1; ++spin)
      if (spin > 256) std::this_thread::yield();
  }

 private:
  void post() {  // the next pass (or the end) is there: spinning workers see the counter, sleeping ones the notification
    {
      std::lock_guard<std::mutex> lk(m_);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
  }
  void work(int t) {
    uint64_t seen = 0;
    for (;;) {
      // a short spin (the serial sections between passes are tens of microseconds), then sleep: a worker that kept spinning
      // would burn the CPU share the caller's serial section needs on a busy host
      for (int spin = 0; spin < 4000 && gen_.load(std::memory_order_acquire) == seen; ++spin) {
      }
      if (gen_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
      }
      ++seen;
      if (!fn_) return;
      fn_(arg_, t);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  const int T_;
  std::vector<std::thread> th_;
  void (*fn_)(void*, int) = nullptr;
  void* arg_ = nullptr;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> done_{0};
  std::mutex m_;
  std::condition_variable cv_;
};
thread_local PassPool* tl_pool = nullptr;
struct PoolScope {  // the pool of an entry point, for its duration
  PassPool pool;
  explicit PoolScope(int T) : pool(T) { tl_pool = T > 1 ? &pool : nullptr; }
  ~PoolScope() { tl_pool = nullptr; }
};

// Runs fn(t) for t in [0, T) on T threads (T == 1: inline): on the entry point's pool when it has one of that size, else on threads of its own.
template <typename F> void parallel_for(int T, F fn) {
  if (T <= 1) { fn(0); return; }
  if (tl_pool && tl_pool->size() == T) { tl_pool->run(fn); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)T);
  for (int t = 0; t < T; ++t) th.emplace_back(fn, t);
  for (auto& x : th) x.join();
}

// host threads for the plan passes: they are memory-bound and start their threads several times over, so beyond 16
// threads they get slower (3.6 / 3.7 / 5.6 / 10 / 20 ms at 8 / 16 / 32 / 64 / 128 threads on the Reddit-scale graph)
// (measured again with the call-local pool and thread-private counters: 32 / 64 threads still lose -- Reddit-scale plan build 3.5 ms at 16,
// 16-17 ms at 32 / 64: the per-thread boundary histograms and the pool's hand-overs grow with the thread count)
int plan_threads(int64_t W) { return W < 4096 ? 1 : std::min(hcspmm::host_threads(), 16); }

int compute_layout(const int32_t* rowptr, int64_t N, const int32_t* bp, const int32_t* ht, const Resolved& rp,
                   bool slicing, Layout* out) {
  const int64_t W = (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H;
  const int T = plan_threads(W);
  std::vector<Layout> part((size_t)T);
  std::vector<int> bad((size_t)T, 0);
  parallel_for(T, [&](int t) {  // counts are sums over windows: any split gives the same totals
    Layout L;
    for (int64_t w = W * t / T; w < W * (t + 1) / T; ++w) {
      const int64_t r0 = w * HCSPMM_BLK_H, r1 = std::min<int64_t>(r0 + HCSPMM_BLK_H, N);
      const int64_t nnz = (int64_t)rowptr[r1] - rowptr[r0];
      if (ht[w] != 0 && nnz > 0) {
        if (bp[w] <= 0) { bad[(size_t)t] = 1; return; }
        L.n_dense++;
        const int64_t K = (int64_t)bp[w] * HCSPMM_BLK_W;
        if (K <= HCSPMM_COMPACT_K) L.n_compact++;
        else if (K <= HCSPMM_COMPACT2_K) L.n_compact2++;
        else L.dense_pack_words += K + (K / 4) * 2;  // U[K] + one 64-bit mask per 4 columns
        L.nnz_dense += nnz;
        L.dense_k_sum += K;
        L.max_dense_k = std::max<int32_t>(L.max_dense_k, (int32_t)K);
      } else {
        L.n_sparse_windows++;
        for (int64_t r = r0; r < r1; ++r) {
          const int64_t d = (int64_t)rowptr[r + 1] - rowptr[r];
          if (slicing && d > rp.slice_threshold) {
            const int64_t pcs = pieces_bound(d, rp);
            L.n_slice_tasks += pcs;
            L.n_partials += pcs;
            L.n_split_rows++;
            L.n_sliced_rows++;
            L.nnz_sliced += d;
          } else if (d > rp.split_threshold) {
            const int64_t segs = (d + rp.segment_len - 1) / rp.segment_len;
            L.n_tasks += segs;
            L.n_partials += segs;
            L.n_split_rows++;
          } else {
            L.n_tasks++;
          }
        }
        L.nnz_sparse += nnz;
      }
    }
    part[(size_t)t] = L;
  });
  Layout L;
  for (int t = 0; t < T; ++t) {
    if (bad[(size_t)t]) return HCSPMM_EINVAL;
    const Layout& p = part[(size_t)t];
    L.n_tasks += p.n_tasks; L.n_dense += p.n_dense; L.n_compact += p.n_compact; L.n_compact2 += p.n_compact2;
    L.n_split_rows += p.n_split_rows; L.n_partials += p.n_partials; L.dense_pack_words += p.dense_pack_words;
    L.nnz_sparse += p.nnz_sparse; L.nnz_dense += p.nnz_dense; L.dense_k_sum += p.dense_k_sum;
    L.n_sparse_windows += p.n_sparse_windows;
    L.n_slice_tasks += p.n_slice_tasks; L.n_sliced_rows += p.n_sliced_rows; L.nnz_sliced += p.nnz_sliced;
    L.max_dense_k = std::max(L.max_dense_k, p.max_dense_k);
  }
  if (slicing) L.n_slice_tasks += (int64_t)kSlicePad * rp.n_slices;  // every slice list is padded to a multiple of kSlicePad
  L.off_tasks = HCSPMM_PLAN_HEADER_WORDS;
  L.off_dense_index = align4(L.off_tasks + 4 * L.n_tasks);
  L.off_dense_pack = align4(L.off_dense_index + 4 * L.n_dense);
  L.off_compact2 = (L.off_dense_pack + L.dense_pack_words + 63) & ~int64_t(63);
  L.off_compact = L.off_compact2 + HCSPMM_COMPACT2_WORDS * L.n_compact2;
  L.off_fixups = align4(L.off_compact + HCSPMM_COMPACT_WORDS * L.n_compact);
  L.off_sparse_windows = align4(L.off_fixups + 4 * L.n_split_rows);
  L.off_slice_table = align4(L.off_sparse_windows + L.n_sparse_windows);
  L.off_slice_tasks = slicing ? align4(L.off_slice_table + rp.n_slices + 1) : L.off_slice_table;
  L.total = align4(L.off_slice_tasks + 4 * (slicing ? L.n_slice_tasks : 0));
  if (L.total > INT32_MAX) return HCSPMM_ERANGE;
  *out = L;
  return HCSPMM_OK;
}

}  // namespace

extern "C" int hcspmm_plan_words(const int32_t* rowptr, int64_t N, int64_t E, const int32_t* bp, const int32_t* ht,
                                 const hcspmm_plan_params* params, int64_t* words_out) {
  if (!rowptr || !words_out || N < 0 || E < 0) return HCSPMM_EINVAL;
  if (N > 0 && (!bp || !ht)) return HCSPMM_EINVAL;
  // whether rows get sliced is decided where the column ids are known (hcspmm_plan_build): unless the parameters
  // settle it, size the tensor for either outcome
  const Resolved rp = resolve(params);
  PoolScope scope(plan_threads((N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H));
  Layout L, Ls;
  int rc = HCSPMM_OK;
  int64_t total = 0;
  if (rp.slice_mode != kSliceOn) {
    rc = compute_layout(rowptr, N, bp, ht, rp, false, &L);
    if (rc != HCSPMM_OK) return rc;
    total = L.total;
  }
  if (rp.slice_mode != kSliceOff) {
    rc = compute_layout(rowptr, N, bp, ht, rp, true, &Ls);
    if (rc != HCSPMM_OK) return rc;
    total = std::max(total, Ls.total);
  }
  *words_out = total;
  return HCSPMM_OK;
}

// power-of-two length class: 0 -> 0, 1 -> 1, 2 -> 2, 3..4 -> 3, 5..8 -> 4, 9..16 -> 5, 17..32 -> 6, ...
static inline int length_class(int32_t len) { return len <= 1 ? (len > 0 ? 1 : 0) : 33 - __builtin_clz((uint32_t)len - 1u); }

// fingerprint of (rowptr, col) + range check of col against [0, M): one parallel pass over both arrays
static int fingerprint_and_check(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int64_t M, uint64_t* out) {
  if (rowptr[0] != 0 || rowptr[N] != E) return HCSPMM_EINVAL;
  const int T = (N + E < (1 << 18)) ? 1 : plan_threads(1 << 20);  // (the entry point's pool when it has one)
  std::vector<uint64_t> part((size_t)T, 0);
  std::vector<int> bad((size_t)T, 0);
  parallel_for(T, [&](int t) {
    uint64_t acc = 0;
    uint32_t over = 0;
    for (int64_t r = (N + 1) * t / T; r < (N + 1) * (t + 1) / T; ++r) acc += hcspmm::fp_term_rowptr((uint64_t)r, rowptr[r]);
    for (int64_t r = std::max<int64_t>(1, (N + 1) * t / T); r < (N + 1) * (t + 1) / T; ++r) over |= (uint32_t)(rowptr[r] < rowptr[r - 1]);
    for (int64_t e = E * t / T; e < E * (t + 1) / T; ++e) {
      acc += hcspmm::fp_term_col((uint64_t)e, col[e]);
      over |= (uint32_t)((uint32_t)col[e] >= (uint32_t)M);
    }
    part[(size_t)t] = acc;
    bad[(size_t)t] = over != 0;
  });
  uint64_t f = hcspmm::fp_seed(N, E);
  for (int t = 0; t < T; ++t) {
    if (bad[(size_t)t]) return HCSPMM_EINVAL;
    f += part[(size_t)t];
  }
  *out = f;
  return HCSPMM_OK;
}

extern "C" int hcspmm_graph_fingerprint_host(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, uint64_t* out) {
  if (!rowptr || !out || N < 0 || E < 0 || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  return fingerprint_and_check(rowptr, col, N, E, (int64_t)INT32_MAX + 1, out);  // any non-negative id passes
}

extern "C" int hcspmm_plan_build(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int64_t M,
                                 const int32_t* bp, const int32_t* e2c, const int32_t* ht,
                                 const hcspmm_plan_params* params, int32_t* plan, int64_t words) {
  if (!rowptr || !plan || N < 0 || E < 0) return HCSPMM_EINVAL;
  if (N > 0 && (!bp || !ht)) return HCSPMM_EINVAL;
  if (E > 0 && (!col || !e2c)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  if (M <= 0) M = N;
  if (M > INT32_MAX) return HCSPMM_ERANGE;
  PoolScope scope(plan_threads((N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H));
  // HCSPMM_PLAN_DEBUG=1: the phases' wall times on stderr
  static const bool dbg = std::getenv("HCSPMM_PLAN_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp[8] = {now(), 0, 0, 0, 0, 0, 0, 0};
  uint64_t fingerprint = 0;
  {
    const int frc = fingerprint_and_check(rowptr, col, N, E, M, &fingerprint);
    if (frc != HCSPMM_OK) return frc;
  }
  tp[1] = now();
  const Resolved rp = resolve(params);
  Layout L;
  int rc = HCSPMM_OK;
  // XCD-affine slices: asked for, or (automatic) worth it -- X spans many L2s and the long rows hold a real share
  bool slicing = rp.slice_mode == kSliceOn;
  if (rp.slice_mode != kSliceOff) {
    rc = compute_layout(rowptr, N, bp, ht, rp, true, &L);
    if (rc != HCSPMM_OK) return rc;
    if (rp.slice_mode == kSliceAuto)
      slicing = M >= kSliceAutoMinColumns && L.nnz_sliced * 20 >= L.nnz_sparse && L.nnz_sliced > 0 &&
                (M >= kSliceAutoAlwaysColumns || L.nnz_sparse >= kSliceAutoMinEntries);
    if (L.n_sliced_rows == 0) slicing = false;
  }
  if (!slicing) {
    rc = compute_layout(rowptr, N, bp, ht, rp, false, &L);
    if (rc != HCSPMM_OK) return rc;
  }
  if (words < L.total) return HCSPMM_EINVAL;
  const int64_t W = (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H;
  const int S = slicing ? rp.n_slices : 0;
  tp[2] = now();

  // Threads work on contiguous window ranges; every output position follows from per-thread counts in
  // range order, so the blob is identical for any thread count.
  const int T = plan_threads(W);
  std::vector<int64_t> cut((size_t)T + 1);
  for (int t = 0; t <= T; ++t) cut[(size_t)t] = W * t / T;
  {  // zero what the passes below do not overwrite word for word: everything behind the task list (masks are OR-ed
     // in, records and section gaps are padded).  The task list itself -- 16 bytes per row, 78 of the 80 MB of a
     // 4.9 M-row plan -- is written completely, so it is not cleared first.
    const int64_t z0 = L.off_dense_index, zn = L.total - z0;
    std::vector<int64_t> zc((size_t)T + 1);
    for (int t = 0; t <= T; ++t) zc[(size_t)t] = z0 + zn * t / T;
    parallel_for(T, [&](int t) {
      std::memset(plan + zc[(size_t)t], 0, sizeof(int32_t) * (size_t)(zc[(size_t)t + 1] - zc[(size_t)t]));
    });
  }

  // ---- slice boundaries: bounds[s] = first column id of slice s + 1, chosen on a histogram of the sliced rows'
  // entries (at most 65536 buckets) so that every slice holds about the same number of them -- equal work per XCD
  std::vector<int32_t> bounds((size_t)std::max(S - 1, 0), 0);
  if (slicing) {
    int shift = 0;
    while (((M - 1) >> shift) >= 65536) ++shift;
    const int64_t nb = ((M - 1) >> shift) + 1;
    std::vector<std::vector<int64_t>> hist((size_t)T, std::vector<int64_t>((size_t)nb, 0));
    parallel_for(T, [&](int t) {
      int64_t* h = hist[(size_t)t].data();
      for (int64_t w = cut[(size_t)t]; w < cut[(size_t)t + 1]; ++w) {
        const int64_t r0 = w * HCSPMM_BLK_H, r1 = std::min<int64_t>(r0 + HCSPMM_BLK_H, N);
        if (ht[w] != 0 && rowptr[r1] > rowptr[r0]) continue;
        for (int64_t r = r0; r < r1; ++r) {
          if ((int64_t)rowptr[r + 1] - rowptr[r] <= rp.slice_threshold) continue;
          for (int64_t e = rowptr[r]; e < rowptr[r + 1]; e += kSliceSample) h[col[e] >> shift]++;  // (a sample: balance needs no more)
        }
      }
    });
    int64_t total = 0;
    for (int64_t b = 0; b < nb; ++b) {
      for (int t = 1; t < T; ++t) hist[0][(size_t)b] += hist[(size_t)t][(size_t)b];
      total += hist[0][(size_t)b];
    }
    int64_t run = 0, b = 0;
    for (int sidx = 0; sidx < S - 1; ++sidx) {
      const int64_t want = (total * (sidx + 1) + S - 1) / S;
      while (b < nb && run + hist[0][(size_t)b] <= want) run += hist[0][(size_t)b++];
      bounds[(size_t)sidx] = (int32_t)std::min<int64_t>(M, b << shift);
    }
  }

  tp[3] = now();
  // ---- sparse tasks.  Order: by descending power-of-two length class (0, 1, 2, 3-4, 5-8, 9-16, ...), rows
  // ascending inside a class.  Classes keep the lane groups of a wave within 2x of each other and put the
  // heavy work first; row order inside a class keeps the Z stores, the column-index reads and the
  // task reads of neighbouring waves close together in memory (a full sort by length scatters them,
  // which costs a few % on low-degree graphs where X and Z live in HBM, not in the Infinity Cache;
  // merging all rows <= 16 entries into ONE row-ordered class is worse, profiles/r01/ab_merge_short.log).
  // Pass 1 counts per (thread, class); pass 2 writes every task straight to its final slot.
  const int n_cls = length_class(rp.split_threshold) + 1;
  struct Counts {
    std::vector<int64_t> cls;
    std::vector<int64_t> slice_cls;  // [slice][class]
    std::vector<int32_t> cuts;       // S + 1 cut points per sliced row of this thread's range, in walk order (pass 1 -> pass 2)
    int64_t n_fix = 0, n_slots = 0, n_dense = 0, n_sparse_w = 0;
  };
  std::vector<Counts> cnt((size_t)T);
  auto walk = [&](int t, bool first_pass, auto&& on_task, auto&& on_fix, auto&& on_dense, auto&& on_sparse_window, auto&& on_piece) {
    std::vector<int32_t>& row_cuts = cnt[(size_t)t].cuts;
    if (first_pass) row_cuts.clear();
    size_t next_cuts = 0;
    for (int64_t w = cut[(size_t)t]; w < cut[(size_t)t + 1]; ++w) {
      const int64_t r0 = w * HCSPMM_BLK_H, r1 = std::min<int64_t>(r0 + HCSPMM_BLK_H, N);
      const int64_t nnz = (int64_t)rowptr[r1] - rowptr[r0];
      if (ht[w] != 0 && nnz > 0) {
        on_dense(w);
        continue;
      }
      on_sparse_window(w);
      for (int64_t r = r0; r < r1; ++r) {
        const int32_t e0 = rowptr[r];
        const int64_t d = (int64_t)rowptr[r + 1] - e0;
        if (slicing && d > rp.slice_threshold) {
          // a row that falls into one piece needs no partial sum; otherwise its pieces take consecutive slots in CSR order
          if (first_pass) {  // the binary searches are done once: pass 2 reads the cut points back
            row_cuts.resize(next_cuts + (size_t)S + 1);
            slice_cuts(col, e0, d, bounds.data(), S, row_cuts.data() + next_cuts);
          }
          const int32_t* cuts = row_cuts.data() + next_cuts;
          next_cuts += (size_t)S + 1;
          int64_t pcs = 0, k = 0;
          for (int sl = 0; sl < S; ++sl) pcs += ((int64_t)cuts[sl + 1] - cuts[sl] + rp.segment_len - 1) / rp.segment_len;
          const int64_t slot0 = pcs > 1 ? on_fix(r, pcs) : -1;
          for_each_piece(cuts, e0, S, rp.segment_len, [&](int sl, int32_t first, int32_t len) {
            on_piece(sl, (int32_t)r, first, len, (int32_t)(pcs > 1 ? slot0 + k : -1));
            ++k;
          });
        } else if (d > rp.split_threshold) {
          const int64_t segs = (d + rp.segment_len - 1) / rp.segment_len;
          const int64_t slot0 = on_fix(r, segs);
          for (int64_t sgm = 0; sgm < segs; ++sgm) {
            const int64_t b = sgm * rp.segment_len;
            on_task((int32_t)r, (int32_t)(e0 + b), (int32_t)std::min<int64_t>(rp.segment_len, d - b), (int32_t)(slot0 + sgm));
          }
        } else {
          on_task((int32_t)r, e0, (int32_t)d, -1);
        }
      }
    }
  };
  parallel_for(T, [&](int t) {
    // counters private to the thread (its own stack and heap blocks) until the end: the Counts records of neighbouring threads share
    // cache lines, and a counter bumped per row there cost the RD-sized graph several milliseconds of line ping-pong
    std::vector<int64_t> cls((size_t)n_cls, 0), slice_cls((size_t)S * (size_t)n_cls, 0);
    int64_t n_fix = 0, n_slots = 0, n_dense = 0, n_sparse_w = 0;
    walk(t, true, [&](int32_t, int32_t, int32_t len, int32_t) { cls[(size_t)length_class(len)]++; },
         [&](int64_t, int64_t segs) { n_fix++; n_slots += segs; return (int64_t)0; },
         [&](int64_t) { n_dense++; }, [&](int64_t) { n_sparse_w++; },
         [&](int sl, int32_t, int32_t, int32_t len, int32_t) { slice_cls[(size_t)sl * (size_t)n_cls + (size_t)length_class(len)]++; });
    Counts& c = cnt[(size_t)t];
    c.cls.swap(cls);
    c.slice_cls.swap(slice_cls);
    c.n_fix = n_fix;
    c.n_slots = n_slots;
    c.n_dense = n_dense;
    c.n_sparse_w = n_sparse_w;
  });
  // class starts (bucket 0 = longest class), then per-thread offsets inside each class, in range order
  std::vector<int64_t> start((size_t)n_cls + 1, 0);
  for (int c = 0; c < n_cls; ++c) {
    int64_t tot = 0;
    for (int t = 0; t < T; ++t) tot += cnt[(size_t)t].cls[(size_t)c];
    start[(size_t)(n_cls - 1 - c) + 1] = tot;
  }
  for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
  int32_t len_gt[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < 5; ++b) {  // prefix sizes: tasks longer than 16 << b  (class boundaries are powers of two)
    const int c = length_class((16 << b) + 1);  // first class whose members are all > 16 << b
    len_gt[b] = c >= n_cls ? 0 : (int32_t)start[(size_t)(n_cls - c)];
  }
  int64_t n_tiny = 0;
  for (int c = 0; c <= length_class(HCSPMM_TINY_LEN) && c < n_cls; ++c)
    for (int t = 0; t < T; ++t) n_tiny += cnt[(size_t)t].cls[(size_t)c];
  std::vector<std::vector<int64_t>> pos((size_t)T, std::vector<int64_t>((size_t)n_cls));
  std::vector<int64_t> fix_at((size_t)T + 1, 0), slot_at((size_t)T + 1, 0), dense_at((size_t)T + 1, 0), sparse_w_at((size_t)T + 1, 0);
  {
    std::vector<int64_t> run(start.begin(), start.end() - 1);  // next free slot per bucket
    for (int t = 0; t < T; ++t) {
      for (int c = 0; c < n_cls; ++c) {
        pos[(size_t)t][(size_t)c] = run[(size_t)(n_cls - 1 - c)];
        run[(size_t)(n_cls - 1 - c)] += cnt[(size_t)t].cls[(size_t)c];
      }
      fix_at[(size_t)t + 1] = fix_at[(size_t)t] + cnt[(size_t)t].n_fix;
      slot_at[(size_t)t + 1] = slot_at[(size_t)t] + cnt[(size_t)t].n_slots;
      dense_at[(size_t)t + 1] = dense_at[(size_t)t] + cnt[(size_t)t].n_dense;
      sparse_w_at[(size_t)t + 1] = sparse_w_at[(size_t)t] + cnt[(size_t)t].n_sparse_w;
    }
  }
  // (cannot happen: compute_layout counted the same things -- with slices, upper bounds of the first two)
  if (slicing ? (fix_at[(size_t)T] > L.n_split_rows || slot_at[(size_t)T] > L.n_partials)
              : (fix_at[(size_t)T] != L.n_split_rows || slot_at[(size_t)T] != L.n_partials))
    return HCSPMM_EINVAL;
  if (dense_at[(size_t)T] != L.n_dense || start.back() != L.n_tasks || sparse_w_at[(size_t)T] != L.n_sparse_windows)
    return HCSPMM_EINVAL;
  // slice lists: slice s owns descriptors [table[s], table[s+1]) (padded to whole waves), longest class first, and inside a
  // class the threads' shares in range order (= rows ascending)
  std::vector<int64_t> table((size_t)S + 1, 0);
  std::vector<std::vector<int64_t>> spos((size_t)T, std::vector<int64_t>((size_t)S * (size_t)n_cls, 0));
  int64_t slice_xcd_tasks = 0;
  if (slicing) {
    int64_t xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int sl = 0; sl < S; ++sl) {
      int64_t run = table[(size_t)sl];
      for (int c = n_cls - 1; c >= 0; --c)
        for (int t = 0; t < T; ++t) {
          spos[(size_t)t][(size_t)sl * (size_t)n_cls + (size_t)c] = run;
          run += cnt[(size_t)t].slice_cls[(size_t)sl * (size_t)n_cls + (size_t)c];
        }
      const int64_t padded = (run - table[(size_t)sl] + kSlicePad - 1) / kSlicePad * kSlicePad;
      int32_t* pad = plan + L.off_slice_tasks;
      for (int64_t q = run; q < table[(size_t)sl] + padded; ++q) {
        if (q >= L.n_slice_tasks) return HCSPMM_EINVAL;
        pad[4 * q + 0] = -1; pad[4 * q + 1] = 0; pad[4 * q + 2] = 0; pad[4 * q + 3] = -1;
      }
      table[(size_t)sl + 1] = table[(size_t)sl] + padded;
      xcd[sl & 7] += padded;
    }
    if (table[(size_t)S] > L.n_slice_tasks) return HCSPMM_EINVAL;
    for (int x = 0; x < 8; ++x) slice_xcd_tasks = std::max(slice_xcd_tasks, xcd[x]);
    for (int sl = 0; sl <= S; ++sl) plan[L.off_slice_table + sl] = (int32_t)table[(size_t)sl];
  }
  tp[4] = now();
  struct DenseRef { int32_t w, K; };
  std::vector<DenseRef> dense((size_t)L.n_dense);
  int32_t* out = plan + L.off_tasks;
  int32_t* fix = plan + L.off_fixups;
  int32_t* sparse_w = plan + L.off_sparse_windows;
  int32_t* sout = plan + L.off_slice_tasks;
  parallel_for(T, [&](int t) {
    std::vector<int64_t> p = pos[(size_t)t];    // thread-private copies: the cursors are bumped per task, and the per-thread vectors
    std::vector<int64_t> sp = spos[(size_t)t];  // the caller allocated sit next to each other in memory
    int64_t n_fix = fix_at[(size_t)t], slot = slot_at[(size_t)t], nd = dense_at[(size_t)t], nsw = sparse_w_at[(size_t)t];
    walk(t, false,
         [&](int32_t row, int32_t e0, int32_t len, int32_t slot_id) {
           const int64_t q = p[(size_t)length_class(len)]++;
           if (len <= HCSPMM_TINY_LEN) {  // classes 2, 1, 0: the tail of the list
             out[4 * q + 0] = slot_id < 0 ? row : -(slot_id + 1);
             out[4 * q + 1] = len >= 1 ? col[e0] : -1;
             out[4 * q + 2] = len;
             out[4 * q + 3] = len >= 2 ? col[e0 + 1] : -1;
           } else {
             out[4 * q + 0] = row;
             out[4 * q + 1] = e0;
             out[4 * q + 2] = len;
             out[4 * q + 3] = slot_id;
           }
         },
         [&](int64_t r, int64_t segs) {
           fix[4 * n_fix + 0] = (int32_t)r;
           fix[4 * n_fix + 1] = (int32_t)slot;
           fix[4 * n_fix + 2] = (int32_t)segs;
           ++n_fix;
           const int64_t s0 = slot;
           slot += segs;
           return s0;
         },
         [&](int64_t w) { dense[(size_t)nd++] = DenseRef{(int32_t)w, bp[w] * HCSPMM_BLK_W}; },
         [&](int64_t w) { sparse_w[nsw++] = (int32_t)w; },
         [&](int sl, int32_t row, int32_t first, int32_t len, int32_t slot_id) {
           const int64_t q = sp[(size_t)sl * (size_t)n_cls + (size_t)length_class(len)]++;
           sout[4 * q + 0] = row;
           sout[4 * q + 1] = first;
           sout[4 * q + 2] = len;
           sout[4 * q + 3] = slot_id;
         });
  });

  tp[5] = now();
  // ---- dense windows: widest first (stable, so window order inside a width); pack U and the MFMA lane masks
  std::stable_sort(dense.begin(), dense.end(), [](const DenseRef& a, const DenseRef& b) { return a.K > b.K; });
  int32_t* dindex = plan + L.off_dense_index;
  int32_t* dpack = plan + L.off_dense_pack;
  // sorted by K: regular windows, then the double-record ones, then the compact ones
  const int64_t n_reg = L.n_dense - L.n_compact - L.n_compact2, n_c2 = L.n_compact2;
  std::vector<int64_t> pack_at((size_t)n_reg + 1, 0);
  for (int64_t i = 0; i < n_reg; ++i) pack_at[(size_t)i + 1] = pack_at[(size_t)i] + dense[(size_t)i].K + (dense[(size_t)i].K / 4) * 2;
  std::vector<int64_t> uniq_part((size_t)T, 0);
  std::vector<int> bad((size_t)T, 0);
  parallel_for(T, [&](int t) {
    int64_t uniq_total = 0;
    for (int64_t i = L.n_dense * t / T; i < L.n_dense * (t + 1) / T; ++i) {
      const int64_t w = dense[(size_t)i].w;
      const int32_t K = dense[(size_t)i].K, K4 = K / 4;
      const int64_t r0 = w * HCSPMM_BLK_H, r1 = std::min<int64_t>(r0 + HCSPMM_BLK_H, N);
      const int kind = i < n_reg ? 0 : (i < n_reg + n_c2 ? 2 : 1);  // 0 regular, 2 double record, 1 compact record
      if (kind != (K <= HCSPMM_COMPACT_K ? 1 : (K <= HCSPMM_COMPACT2_K ? 2 : 0))) { bad[(size_t)t] = 1; return; }
      const int32_t kmax = kind == 1 ? HCSPMM_COMPACT_K : (kind == 2 ? HCSPMM_COMPACT2_K : K);  // U slots
      const int64_t off = kind == 1 ? HCSPMM_COMPACT_WORDS * (i - n_reg - n_c2)
                                    : (kind == 2 ? HCSPMM_COMPACT2_WORDS * (i - n_reg) : pack_at[(size_t)i]);
      int32_t* rec = plan + (kind == 1 ? L.off_compact : L.off_compact2) + off;
      int32_t* U = kind ? rec + 2 : dpack + off;
      // little-endian halves of the 64-bit masks
      uint32_t* masks = reinterpret_cast<uint32_t*>(U + kmax);
      if (kind) {
        rec[0] = (int32_t)w;
        rec[1] = K4;
      }
      for (int32_t k = 0; k < kmax; ++k) U[k] = -1;
      for (int64_t r = r0; r < r1; ++r) {
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
          const int32_t c = e2c[e];
          if (c < 0 || c >= K) { bad[(size_t)t] = 1; return; }
          U[c] = col[e];
          const int lane = 16 * (c & 3) + (int)(r - r0);
          masks[(c >> 2) * 2 + (lane >> 5)] |= 1u << (lane & 31);
        }
      }
      for (int32_t k = 0; k < K; ++k) uniq_total += U[k] >= 0;
      dindex[4 * i + 0] = (int32_t)w;
      dindex[4 * i + 1] = (int32_t)off;
      dindex[4 * i + 2] = K4;
      dindex[4 * i + 3] = kind;  // 1 / 2: offset is relative to that compact section (the kernel does not read the entry)
    }
    uniq_part[(size_t)t] = uniq_total;
  });
  int64_t uniq_total = 0;
  for (int t = 0; t < T; ++t) {
    if (bad[(size_t)t]) return HCSPMM_EINVAL;
    uniq_total += uniq_part[(size_t)t];
  }

  hcspmm_plan_header h;
  std::memset(&h, 0, sizeof(h));
  h.magic = HCSPMM_PLAN_MAGIC;
  h.version = HCSPMM_PLAN_VERSION;
  h.total_words = (int32_t)L.total;
  h.num_nodes = (int32_t)N;
  h.num_edges = (int32_t)E;
  h.num_windows = (int32_t)W;
  h.split_threshold = rp.split_threshold;
  h.segment_len = rp.segment_len;
  h.n_tasks = (int32_t)L.n_tasks;
  h.n_dense = (int32_t)L.n_dense;
  h.n_split_rows = (int32_t)fix_at[(size_t)T];
  h.n_partials = (int32_t)slot_at[(size_t)T];
  h.off_tasks = (int32_t)L.off_tasks;
  h.off_dense_index = (int32_t)L.off_dense_index;
  h.off_dense_pack = (int32_t)L.off_dense_pack;
  h.off_fixups = (int32_t)L.off_fixups;
  h.nnz_sparse = (int32_t)L.nnz_sparse;
  h.nnz_dense = (int32_t)L.nnz_dense;
  h.uniq_dense = (int32_t)uniq_total;
  h.max_dense_k = L.max_dense_k;
  for (int b = 0; b < 5; ++b) h.n_len_gt[b] = len_gt[b];
  h.n_tiny = (int32_t)n_tiny;
  h.n_dense_compact = (int32_t)L.n_compact;
  h.off_dense_compact = (int32_t)L.off_compact;
  h.n_dense_compact2 = (int32_t)L.n_compact2;
  h.off_dense_compact2 = (int32_t)L.off_compact2;
  h.num_columns = (int32_t)M;
  h.n_sparse_windows = (int32_t)L.n_sparse_windows;
  h.off_sparse_windows = (int32_t)L.off_sparse_windows;
  h.fingerprint_lo = (uint32_t)(fingerprint & 0xffffffffull);
  h.fingerprint_hi = (uint32_t)(fingerprint >> 32);
  h.dense_k_sum = (int32_t)std::min<int64_t>(L.dense_k_sum, INT32_MAX);
  h.flags = rp.fuse_in_launch < 0 ? HCSPMM_PLAN_FUSE_NEVER : (rp.fuse_in_launch >= 2 ? HCSPMM_PLAN_FUSE_ROWS : (rp.fuse_in_launch == 1 ? HCSPMM_PLAN_FUSE_IN_LAUNCH : 0));
  h.n_slices = S;
  h.slice_threshold = slicing ? rp.slice_threshold : 0;
  h.off_slice_table = (int32_t)L.off_slice_table;
  h.off_slice_tasks = (int32_t)L.off_slice_tasks;
  h.n_slice_tasks = slicing ? (int32_t)table[(size_t)S] : 0;
  h.slice_xcd_tasks = (int32_t)slice_xcd_tasks;
  h.nnz_sliced = slicing ? (int32_t)L.nnz_sliced : 0;
  h.n_sliced_rows = slicing ? (int32_t)L.n_sliced_rows : 0;
  h.panel_cols = rp.panel_cols;
  static_assert(sizeof(hcspmm_plan_header) == HCSPMM_PLAN_HEADER_WORDS * 4, "header size");
  std::memcpy(plan, &h, sizeof(h));
  if (dbg) {
    tp[6] = now();
    std::fprintf(stderr, "plan_build: T=%d: fingerprint + range check %.2f | layout %.2f | clear + slice boundaries %.2f | task pass 1 (count) %.2f | "
                         "task pass 2 (write) %.2f | dense packs %.2f | total %.2f ms\n",
                 T, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], tp[6] - tp[0]);
  }
  return HCSPMM_OK;
}

extern "C" int hcspmm_plan_check(const hcspmm_plan_header* h, int64_t N, int64_t E, int64_t words_available) {
  if (!h) return HCSPMM_EINVAL;
  if (h->magic != HCSPMM_PLAN_MAGIC || h->version != HCSPMM_PLAN_VERSION) return HCSPMM_EPLAN;
  if (h->num_nodes != N || h->num_edges != E) return HCSPMM_EPLAN;
  if (h->num_windows != (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H) return HCSPMM_EPLAN;
  if (h->num_columns < 0 || (h->num_columns == 0 && N > 0)) return HCSPMM_EPLAN;  // (the plan of an empty graph gathers from nothing)
  if (h->n_tasks < 0 || h->n_dense < 0 || h->n_split_rows < 0 || h->n_partials < 0) return HCSPMM_EPLAN;
  if (h->n_tiny < 0 || h->n_tiny > h->n_tasks) return HCSPMM_EPLAN;
  if (h->n_dense_compact < 0 || h->n_dense_compact2 < 0 ||
      (int64_t)h->n_dense_compact + h->n_dense_compact2 > h->n_dense)
    return HCSPMM_EPLAN;
  if (h->n_sparse_windows < 0 || (int64_t)h->n_sparse_windows + h->n_dense != h->num_windows) return HCSPMM_EPLAN;
  if (h->n_partials < 2 * (int64_t)h->n_split_rows || h->split_threshold <= 0 || h->segment_len <= 0) return HCSPMM_EPLAN;
  // the wide-task prefixes index the non-tiny part of the task list, longest first
  for (int b = 0; b < 5; ++b) {
    if (h->n_len_gt[b] < 0 || h->n_len_gt[b] > h->n_tasks - h->n_tiny) return HCSPMM_EPLAN;
    if (b > 0 && h->n_len_gt[b] > h->n_len_gt[b - 1]) return HCSPMM_EPLAN;
  }
  // every section starts where the layout says and ends inside the blob, in this order
  const int64_t n_reg = (int64_t)h->n_dense - h->n_dense_compact - h->n_dense_compact2;
  if (h->off_tasks != HCSPMM_PLAN_HEADER_WORDS) return HCSPMM_EPLAN;
  if (h->off_dense_index < h->off_tasks + 4 * (int64_t)h->n_tasks || (h->off_dense_index & 3)) return HCSPMM_EPLAN;
  if (h->off_dense_pack < h->off_dense_index + 4 * (int64_t)h->n_dense || (h->off_dense_pack & 3)) return HCSPMM_EPLAN;
  // a regular pack holds at least 88 + 44 words (K > 80)
  if (h->off_dense_compact2 < h->off_dense_pack + 132 * n_reg || (h->off_dense_compact2 & 63)) return HCSPMM_EPLAN;
  if (h->off_dense_compact != h->off_dense_compact2 + (int64_t)HCSPMM_COMPACT2_WORDS * h->n_dense_compact2) return HCSPMM_EPLAN;
  if (h->off_fixups < h->off_dense_compact + (int64_t)HCSPMM_COMPACT_WORDS * h->n_dense_compact || (h->off_fixups & 3))
    return HCSPMM_EPLAN;
  if (h->off_sparse_windows < h->off_fixups + 4 * (int64_t)h->n_split_rows || (h->off_sparse_windows & 3)) return HCSPMM_EPLAN;
  if (h->total_words < h->off_sparse_windows + (int64_t)h->n_sparse_windows) return HCSPMM_EPLAN;
  if (h->n_slices < 0 || h->n_slices > 64 || (h->n_slices & 7) || h->n_slice_tasks < 0 || h->slice_xcd_tasks < 0) return HCSPMM_EPLAN;
  if (h->n_slices > 0) {
    if (h->slice_threshold <= 0 || (h->n_slice_tasks % 64) || (h->slice_xcd_tasks % 64) || h->slice_xcd_tasks > h->n_slice_tasks)
      return HCSPMM_EPLAN;
    if (h->off_slice_table < h->off_sparse_windows + (int64_t)h->n_sparse_windows || (h->off_slice_table & 3)) return HCSPMM_EPLAN;
    if (h->off_slice_tasks < h->off_slice_table + (int64_t)h->n_slices + 1 || (h->off_slice_tasks & 3)) return HCSPMM_EPLAN;
    if (h->total_words < h->off_slice_tasks + 4 * (int64_t)h->n_slice_tasks) return HCSPMM_EPLAN;
    if (h->nnz_sliced < 0 || h->nnz_sliced > h->nnz_sparse) return HCSPMM_EPLAN;
  } else if (h->n_slice_tasks != 0 || h->slice_xcd_tasks != 0) {
    return HCSPMM_EPLAN;
  }
  if (words_available > 0 && h->total_words > words_available) return HCSPMM_EPLAN;
  if ((int64_t)h->nnz_sparse + h->nnz_dense != E) return HCSPMM_EPLAN;
  return HCSPMM_OK;
}

extern "C" size_t hcspmm_workspace_bytes(const hcspmm_plan_header* h, int D) {
  if (!h || D <= 0) return 0;
  return (size_t)h->n_partials * (size_t)D * sizeof(float);
}
