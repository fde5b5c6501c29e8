// loi_fast_host.cpp -- hcspmm_loi_reorder_fast: a RELAXED, parallel form of the LOI layout reorder.
//
// The exact reorder (loi_host.cpp; reorder_plus_new_direct, LOI.cpp:660-805) is a strictly sequential greedy whose
// cost is the sum of the lengths of every column list a group touches: 6.7 s for a Reddit-scale graph on the GPU
// box's host (profiles/r02/loi_timing.log), more than a 200-epoch training run takes on MI355X.  This variant keeps
// the reference's group growth -- seed = next unplaced non-empty row, up to 15 additions of the candidate maximising
// (float)(ones + deg v) / (cols + deg v - shared v), first discovered winning ties (LOI.cpp:726-727, :775-781), the
// same output order (LOI.cpp:873-891) -- and relaxes two things, which is why its permutation is NOT the reference's:
//  * list cap: one walk of a column's row list looks at no more than `list_cap` rows behind the list's leading run
//    of placed rows (hub columns say little about which rows belong together and are most of the exact run time);
//  * rounds: `batch` seeds are grown concurrently against the placement state of the round's start; a row wanted by
//    several groups of a round goes to the earliest seed (deterministic reservations: an atomic minimum per row),
//    the others keep what they won.
// The result depends on (graph, batch, list_cap) only -- never on the number of threads or on timing.  With
// batch = 1 and list_cap < 0 neither relaxation is active and the permutation IS hcspmm_loi_reorder's, bit for bit
// (tests/test_host_cpu.py checks that against the fixtures made by the reference's own LOI.cpp).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include <sched.h>

#include <fstream>
#include <string>

#include "hcspmm.h"
#include "host_util.h"
#include "loi_scan.h"

namespace {

// Sense-reversing barrier: a short spin, then yield (the GPU box gives a job a CPU share, not whole cores: a waiter that
// only spins burns the quota its peers need -- profiles/r02/loi_timing.log).
class Barrier {
 public:
  explicit Barrier(int n) : n_(n), count_(0), sense_(0) {}
  void wait() {
    const int s = sense_.load(std::memory_order_acquire);
    if (count_.fetch_add(1, std::memory_order_acq_rel) == n_ - 1) {
      count_.store(0, std::memory_order_relaxed);
      sense_.store(s ^ 1, std::memory_order_release);
      return;
    }
    for (int spin = 0; sense_.load(std::memory_order_acquire) == s; ++spin)
      if (spin > 512) std::this_thread::yield();
  }

 private:
  const int n_;
  std::atomic<int> count_, sense_;
};

// Open-addressing map int32 key (>= 0) -> int32 value, cleared by walking the slots it filled (groups are small; the
// exact code's per-vertex stamp arrays would be 8 bytes x N per thread and a cache miss per probe).
struct SmallMap {
  std::vector<int32_t> key, val;
  std::vector<uint32_t> filled;
  uint32_t mask = 0;
  int shift = 32;
  void reset(uint32_t cap_pow2) {
    key.assign(cap_pow2, -1);
    val.assign(cap_pow2, 0);
    filled.clear();
    mask = cap_pow2 - 1;
    shift = 32 - __builtin_ctz(cap_pow2);
  }
  void clear() {
    if (filled.size() * 4 > key.size()) {
      std::fill(key.begin(), key.end(), -1);
    } else {
      for (uint32_t s : filled) key[s] = -1;
    }
    filled.clear();
  }
  void grow() {
    std::vector<int32_t> ok, ov;
    ok.swap(key);
    ov.swap(val);
    std::vector<uint32_t> of;
    of.swap(filled);
    reset((uint32_t)ok.size() * 2);
    for (uint32_t s : of) *slot(ok[s], nullptr) = ov[s];
  }
  // address of the value of k, inserted (value unset) when absent; *fresh tells which
  int32_t* slot(int32_t k, bool* fresh) {
    if ((filled.size() + 1) * 2 > key.size()) grow();
    uint32_t s = ((uint32_t)k * 2654435761u) >> shift;
    for (;; s = (s + 1) & mask) {
      if (key[s] == k) {
        if (fresh) *fresh = false;
        return &val[s];
      }
      if (key[s] < 0) {
        key[s] = k;
        filled.push_back(s);
        if (fresh) *fresh = true;
        return &val[s];
      }
    }
  }
};

// Shared, read-mostly state of a run.  Everything a group's growth touches per column sits in ONE 8-byte record and
// the placement state is a bitmap (N / 8 bytes: cache-resident where a byte per vertex is not) -- the run is bound by
// cache misses, not by instructions.
struct ColMeta {
  int32_t begin;  // first entry of the column's row list in col_in (the list ends where the next column's begins)
  int32_t cur;    // entries before this one are placed rows (only ever advanced; every thread computes the same value)
};
struct Shared {
  const int32_t *rowptr, *col, *col_in;
  ColMeta* meta;           // [N + 1]
  const uint64_t* placed;  // bit v: vertex v is in a group -- the state of the round's START (written between rounds only)
  const uint8_t* deg8;     // min(out-degree, 255)
  int32_t list_cap;
};

inline int32_t load_relaxed(const int32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
inline void store_relaxed(int32_t* p, int32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
inline bool is_placed(const uint64_t* bits, int32_t v) { return (bits[(uint32_t)v >> 6] >> ((uint32_t)v & 63)) & 1; }
inline int32_t degree(const Shared& g, int32_t v) {
  const int32_t d = g.deg8[v];
  return d < 255 ? d : g.rowptr[v + 1] - g.rowptr[v];
}
inline void prefetch(const void* p) { __builtin_prefetch(p, 0, 3); }

// One group being grown.  The growth is a chain of dependent cache misses (row of the latest member -> its columns'
// records -> their row lists), so a thread keeps several groups in flight and advances them one stage at a time, each
// stage ending with the prefetches the next one needs: the misses of different groups overlap.
struct Grower {
  enum Stage { kIdle, kRow, kColumns, kMeta, kWalk };
  Stage stage = kIdle;
  std::vector<int32_t> cand, cand_deg, cand_shared, cand_alive, resi;
  SmallMap where, cols;  // row -> position among the candidates; the group's columns (value unused)
  int32_t out[16];     // members, seed first (copied to the shared array when the group is complete: neighbouring groups are
                       // grown by different threads at the same time and would otherwise share cache lines)
  int64_t index = -1;  // of the group within the round
  int n = 0, step = 0;
  int32_t latest = -1, ones = 0, ncols = 0, row_begin = 0, row_end = 0;
  Grower() {
    where.reset(1024);
    cols.reset(256);
  }

  void start(const Shared& g, int32_t seed, int64_t index_) {
    cand.clear();
    cand_deg.clear();
    cand_shared.clear();
    cand_alive.clear();
    where.clear();
    cols.clear();
    resi.clear();
    index = index_;
    n = 0;
    step = 0;
    ones = 0;
    ncols = 0;
    // the seed sits among the candidates as a dead entry: the frozen state does not know it has been taken
    bool fresh;
    *where.slot(seed, &fresh) = 0;
    cand.push_back(seed);
    cand_deg.push_back(0);
    cand_shared.push_back(0);
    cand_alive.push_back(0);
    latest = seed;
    out[n++] = seed;
    prefetch(&g.rowptr[seed]);
    stage = kRow;
  }

  // Where the walk of column c's list begins.  A list longer than the cap is a hub column's: it says little about which
  // rows belong together and its rows would swamp the pricing scan (15 passes over every candidate), so only 16 of its
  // unplaced rows are looked at -- enough for the children of one hub to fill a group -- and WHICH 16 depends on the
  // seed, or every group of a round would see the same few rows through the hub and lose them to the earliest seed.
  int32_t window_start(const Shared& g, int32_t c, int32_t cur, int32_t end) const {
    if (g.list_cap < 0 || end - g.meta[c].begin <= g.list_cap) return cur;
    const int32_t room = end - cur - std::min(16, g.list_cap);
    if (room <= 0) return cur;
    return cur + (int32_t)((((uint32_t)out[0] * 2654435761u) ^ ((uint32_t)c * 0x9E3779B1u)) % (uint32_t)(room + 1));
  }

  // one stage; returns false when the group is complete
  bool advance(const Shared& g) {
    bool fresh;
    switch (stage) {
      case kRow: {  // the latest member's row pointers
        row_begin = g.rowptr[latest];
        row_end = g.rowptr[latest + 1];
        prefetch(&g.col[row_begin]);
        stage = kColumns;
        return true;
      }
      case kColumns: {  // its columns join the group; the new ones are walked next
        resi.clear();
        for (int32_t e = row_begin; e < row_end; ++e) {
          const int32_t c = g.col[e];
          cols.slot(c, &fresh);
          if (fresh) ++ncols;
          // (the seed's columns are walked like any residual, duplicates too, as the exact code does.  With a list cap, a member's
          // columns beyond the first 4 * list_cap new ones still count for the profit but are not walked: a row of thousands of
          // entries is a hub row -- it shares a column with everything and groups with nothing -- and walking all of its lists
          // is most of the run time on power-law graphs)
          if ((fresh || n == 1) && (g.list_cap < 0 || (int64_t)resi.size() < 4 * (int64_t)g.list_cap)) {
            resi.push_back(c);
            prefetch(&g.meta[c]);
          }
        }
        ones += row_end - row_begin;
        if (n == 16) return false;  // (complete: the 15th addition needs no further walk)
        stage = kMeta;
        return true;
      }
      case kMeta: {  // where the lists start
        for (int32_t c : resi) {
          const int32_t end = g.meta[c + 1].begin;
          const int32_t j = window_start(g, c, load_relaxed(&g.meta[c].cur), end);
          prefetch(&g.col_in[j]);
          if (end - j > 16) prefetch(&g.col_in[j + 16]);
        }
        stage = kWalk;
        return true;
      }
      case kWalk: {  // candidates through the new columns, then the pick (LOI.cpp:726-727 / :775-781)
        for (int32_t c : resi) {
          const int32_t end = g.meta[c + 1].begin;
          int32_t j = load_relaxed(&g.meta[c].cur);
          const int32_t j0 = j;
          while (j < end && is_placed(g.placed, g.col_in[j])) ++j;  // leading placed rows: skipped for good
          if (j != j0) store_relaxed(&g.meta[c].cur, j);
          const int32_t lim = end - g.meta[c].begin > g.list_cap ? std::min(16, g.list_cap) : g.list_cap;
          j = window_start(g, c, j, end);
          const int32_t stop = g.list_cap < 0 ? end : (int32_t)std::min<int64_t>(end, (int64_t)j + lim);
          for (; j < stop; ++j) {
            const int32_t r = g.col_in[j];
            if (is_placed(g.placed, r)) continue;
            int32_t* pos = where.slot(r, &fresh);
            if (fresh) {
              *pos = (int32_t)cand.size();
              cand.push_back(r);
              cand_deg.push_back(degree(g, r));
              cand_shared.push_back(0);
              cand_alive.push_back(-1);
            }
            cand_shared[(size_t)*pos]++;
          }
        }
        const int32_t base = n == 1 ? ones : ncols;
        const int64_t bp = hcspmm::loi::scan_best(cand_deg.data(), cand_shared.data(), cand_alive.data(), (int64_t)cand.size(), ones, base);
        if (bp < 0) return false;
        latest = cand[(size_t)bp];
        cand_alive[(size_t)bp] = 0;
        out[n++] = latest;
        prefetch(&g.rowptr[latest]);
        stage = kRow;
        return true;
      }
      default: return false;
    }
  }
};

// The CPUs that share a cache level (or a memory node) with the CPU this thread runs on, within the CPUs the process may
// use: "l3" -> /sys/devices/system/cpu/cpuK/cache/index3/shared_cpu_list, "node" -> the NUMA node's cpulist.  The run is
// bound by cache misses and by lines bouncing between cores (reservations, the placement bitmap), so WHERE the threads
// sit matters more than how many there are: on a two-socket host the same 16 threads take 0.2 to 0.9 s depending on
// where the scheduler happens to put them.  Returns false when the topology cannot be read (then nothing is pinned).
bool locality_cpuset(const char* mode, cpu_set_t* out) {
  const int cpu = sched_getcpu();
  if (cpu < 0) return false;
  std::string path;
  if (std::string(mode) == "l3") {
    path = "/sys/devices/system/cpu/cpu" + std::to_string(cpu) + "/cache/index3/shared_cpu_list";
  } else {
    for (int node = 0; node < 64 && path.empty(); ++node) {
      std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
      std::string line;
      if (!f || !std::getline(f, line)) continue;
      // is `cpu` in this list?
      size_t i = 0;
      while (i < line.size()) {
        const int a = std::atoi(line.c_str() + i);
        int b = a;
        size_t j = line.find_first_of(",-", i);
        if (j != std::string::npos && line[j] == '-') {
          b = std::atoi(line.c_str() + j + 1);
          j = line.find(',', j);
        }
        if (cpu >= a && cpu <= b) path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        if (j == std::string::npos) break;
        i = j + 1;
      }
    }
    if (path.empty()) return false;
  }
  std::ifstream f(path);
  std::string line;
  if (!f || !std::getline(f, line)) return false;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
  CPU_ZERO(out);
  int n = 0;
  size_t i = 0;
  while (i < line.size()) {
    const int a = std::atoi(line.c_str() + i);
    int b = a;
    size_t j = line.find_first_of(",-", i);
    if (j != std::string::npos && line[j] == '-') {
      b = std::atoi(line.c_str() + j + 1);
      j = line.find(',', j);
    }
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) {
        CPU_SET(c, out);
        ++n;
      }
    if (j == std::string::npos) break;
    i = j + 1;
  }
  return n > 0;
}

template <class F>
void run_threads(int T, F&& f) {
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(f, t);
  f(0);
  for (auto& x : th) x.join();
}

}  // namespace

extern "C" int hcspmm_loi_reorder_fast(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E,
                                       const hcspmm_loi_fast_params* params, int32_t* perm_out, int32_t* group_sizes_out,
                                       int64_t* n_groups_out) {
  if (N < 0 || E < 0 || !rowptr || (N > 0 && !perm_out) || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  if (!hcspmm::csr_row_pointers_ok(rowptr, N, E)) return HCSPMM_EINVAL;
  hcspmm_loi_fast_params p = {0, 0, 0, 0};
  if (params) p = *params;
  if (p.batch < 0 || p.threads < 0) return HCSPMM_EINVAL;
  const int64_t batch = p.batch > 0 ? p.batch : std::max<int64_t>(1, std::min<int64_t>(2048, N / 2048));
  const int32_t list_cap = p.list_cap == 0 ? 64 : p.list_cap;
  int T = p.threads > 0 ? std::min(p.threads, 256) : std::min(16, hcspmm::host_threads());
  if (batch == 1 || N < 4096) T = 1;
  if (N == 0) {
    if (n_groups_out) *n_groups_out = 0;
    return HCSPMM_OK;
  }

  const bool dbg = std::getenv("HCSPMM_LOI_DEBUG") != nullptr;
  // keep the threads of this call together (see locality_cpuset); the caller's own mask is put back at the end
  cpu_set_t caller_mask, near_mask;
  const char* pin_mode = std::getenv("HCSPMM_LOI_PIN");
  // default: the caller's L3 domain for low-degree graphs (a chain of cache misses per group: SMT siblings and a shared L3 help), its
  // NUMA node for denser ones (hash inserts and pricing scans over thousands of candidates per group want whole cores: the Reddit-scale
  // graph takes 0.71 s on 16 hardware threads of one L3 domain, 0.45-0.5 s on 16 cores of the node)
  if (!pin_mode) pin_mode = E <= 8 * N ? "l3" : "node";
  // (a domain with fewer CPUs than half the threads would crowd them: the L3 domain of a desktop part or of a small VM; then the
  // NUMA node is tried, then nothing is pinned)
  bool pinned = false;
  if (T > 1 && std::string(pin_mode) != "off" && sched_getaffinity(0, sizeof(caller_mask), &caller_mask) == 0) {
    bool have = locality_cpuset(pin_mode, &near_mask) && 2 * CPU_COUNT(&near_mask) >= T;
    if (!have && std::string(pin_mode) == "l3") {
      pin_mode = "node";
      have = locality_cpuset(pin_mode, &near_mask) && 2 * CPU_COUNT(&near_mask) >= T;
    }
    pinned = have && sched_setaffinity(0, sizeof(near_mask), &near_mask) == 0;
  }
  struct Restore {
    bool on;
    cpu_set_t* m;
    ~Restore() {
      if (on) sched_setaffinity(0, sizeof(cpu_set_t), m);
    }
  } restore{pinned, &caller_mask};  // (threads started from here on inherit the mask)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  // ---- rows that reference a column, ascending (the in-CSR of LOI.cpp:826-841), in two levels so that every entry is
  // read a fixed number of times whatever the thread count: (1) each thread takes a range of rows and scatters its
  // (column, row) pairs into buckets of 2^shift consecutive columns -- thread ranges are in row order and a thread
  // writes in row order, so a bucket holds its pairs in row order; (2) each bucket is turned into its columns' lists
  // with counters that fit the L1 cache.
  int shift = 12;
  while (((N - 1) >> shift) >= 4096) ++shift;
  const int64_t n_buckets = ((N - 1) >> shift) + 1;
  std::vector<int32_t> col_in((size_t)E), pair_col((size_t)E), pair_row((size_t)E);
  std::vector<ColMeta> meta((size_t)N + 1);
  std::vector<uint8_t> deg8((size_t)N);
  std::atomic<int> bad{0};
  {
    Barrier bar(T);
    std::vector<int64_t> cnt((size_t)T * (size_t)n_buckets, 0), bucket_begin((size_t)n_buckets + 1, 0);
    std::atomic<int64_t> next_bucket{0};
    run_threads(T, [&](int t) {
      // rows [r0, r1): an equal share of the entries
      const int64_t r0 = std::lower_bound(rowptr, rowptr + N, (int32_t)(E * t / T)) - rowptr;
      const int64_t r1 = t == T - 1 ? N : std::lower_bound(rowptr, rowptr + N, (int32_t)(E * (t + 1) / T)) - rowptr;
      int64_t* my = &cnt[(size_t)t * (size_t)n_buckets];
      for (int64_t r = r0; r < r1; ++r) deg8[(size_t)r] = (uint8_t)std::min(255, rowptr[r + 1] - rowptr[r]);
      for (int64_t e = rowptr[r0]; e < rowptr[r1]; ++e) {
        const int32_t c = col[e];
        if ((uint32_t)c >= (uint32_t)N) {
          bad.store(1, std::memory_order_relaxed);
          continue;
        }
        my[c >> shift]++;
      }
      bar.wait();
      if (bad.load(std::memory_order_relaxed)) return;
      if (t == 0) {  // bucket b starts after the buckets before it; inside it, thread t's pairs follow those of threads < t
        int64_t run = 0;
        for (int64_t b = 0; b < n_buckets; ++b) {
          bucket_begin[(size_t)b] = run;
          for (int u = 0; u < T; ++u) {
            const int64_t k = cnt[(size_t)u * (size_t)n_buckets + (size_t)b];
            cnt[(size_t)u * (size_t)n_buckets + (size_t)b] = run;
            run += k;
          }
        }
        bucket_begin[(size_t)n_buckets] = run;
      }
      bar.wait();
      for (int64_t r = r0; r < r1; ++r)
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
          const int64_t at = my[col[e] >> shift]++;
          pair_col[(size_t)at] = col[e];
          pair_row[(size_t)at] = (int32_t)r;
        }
      bar.wait();
      std::vector<int32_t> local((size_t)1 << shift);
      for (;;) {
        const int64_t b = next_bucket.fetch_add(1, std::memory_order_relaxed);
        if (b >= n_buckets) break;
        const int64_t c0 = b << shift, c1 = std::min<int64_t>(N, c0 + ((int64_t)1 << shift));
        const int64_t p0 = bucket_begin[(size_t)b], p1 = bucket_begin[(size_t)b + 1];
        std::fill(local.begin(), local.begin() + (c1 - c0), 0);
        for (int64_t p = p0; p < p1; ++p) local[(size_t)(pair_col[(size_t)p] - c0)]++;
        int64_t run = p0;
        for (int64_t c = c0; c < c1; ++c) {
          const int32_t k = local[(size_t)(c - c0)];
          meta[(size_t)c] = ColMeta{(int32_t)run, (int32_t)run};
          local[(size_t)(c - c0)] = (int32_t)run;
          run += k;
        }
        for (int64_t p = p0; p < p1; ++p) col_in[(size_t)local[(size_t)(pair_col[(size_t)p] - c0)]++] = pair_row[(size_t)p];
      }
      if (t == 0) meta[(size_t)N] = ColMeta{(int32_t)E, (int32_t)E};
    });
  }
  if (bad.load()) return HCSPMM_EINVAL;
  std::vector<int32_t>().swap(pair_col);
  std::vector<int32_t>().swap(pair_row);

  const double t_incsr = now();
  std::vector<uint64_t> placed(((size_t)N + 63) / 64 + 1, 0);
  std::vector<int32_t> claim((size_t)N, INT32_MAX);
  // members: 16 slots per group, one uninitialised block per round (a zero-filled, growing vector cost page faults and copies)
  std::vector<std::unique_ptr<int32_t[]>> round_members;
  std::vector<int64_t> round_first;  // index of a round's first group
  int32_t* cur_members = nullptr;
  std::vector<int32_t> sizes, seeds((size_t)batch), group_seg, group_round;
  sizes.reserve((size_t)N / 8 + (size_t)batch);
  // The seeds of a round come from `batch` SEGMENTS of the vertex range, one cursor each (the next unplaced non-empty row of the
  // segment): seeds of one round are then far apart in the numbering.  Taking the next `batch` unplaced rows instead put a whole
  // round into a few neighbouring molecules of an already-local graph -- every group of the round wanted the same rows, most lost
  // them (86 K full groups of 420 K on the YeastH-sized collection, and 4 x the run time).  Groups are emitted in ascending seed order
  // (segment by segment, rounds ascending inside a segment), which is the creation order of the exact algorithm; batch = 1 is one
  // segment with one cursor, i.e. exactly that algorithm.
  std::vector<int64_t> cursor((size_t)batch), seg_end((size_t)batch);
  for (int64_t k = 0; k < batch; ++k) {
    cursor[(size_t)k] = N * k / batch;
    seg_end[(size_t)k] = N * (k + 1) / batch;
  }
  Shared g{rowptr, col, col_in.data(), meta.data(), placed.data(), deg8.data(), list_cap};

  int64_t n_seeds = 0, base = 0;  // base: groups created before this round
  std::atomic<int64_t> next{0};
  Barrier bar(T);
  double t_sel = 0, t_grow = 0, t_place = 0;  // (thread 0's view, HCSPMM_LOI_DEBUG)
  std::vector<std::vector<Grower>> growers((size_t)T);
  run_threads(T, [&](int t) {
    std::vector<Grower>& gr = growers[(size_t)t];
    gr.resize(8);
    for (;;) {
      const double ta = t == 0 && dbg ? now() : 0;
      if (t == 0) {
        n_seeds = 0;
        base = (int64_t)sizes.size();
        for (int64_t k = 0; k < batch; ++k) {  // per segment: its next unplaced non-empty row, a bitmap word at a time
          int64_t v = cursor[(size_t)k];
          const int64_t end = seg_end[(size_t)k];
          int64_t found = -1;
          while (v < end) {
            const uint64_t free_bits = ~placed[(size_t)(v >> 6)] & (~(uint64_t)0 << (v & 63));
            if (!free_bits) {
              v = (v & ~(int64_t)63) + 64;
              continue;
            }
            v = (v & ~(int64_t)63) + __builtin_ctzll(free_bits);
            if (v >= end) break;
            if (rowptr[v + 1] > rowptr[v]) {
              found = v;
              break;
            }
            ++v;
          }
          cursor[(size_t)k] = found >= 0 ? found + 1 : end;
          if (found >= 0) {
            seeds[(size_t)n_seeds++] = (int32_t)found;
            group_seg.push_back((int32_t)k);
            group_round.push_back((int32_t)round_members.size());
          }
        }
        if (n_seeds > 0) {
          round_members.emplace_back(new int32_t[(size_t)n_seeds * 16]);
          round_first.push_back(base);
          cur_members = round_members.back().get();
        }
        sizes.resize((size_t)(base + n_seeds));
        next.store(0, std::memory_order_relaxed);
      }
      bar.wait();
      if (n_seeds == 0) break;
      const double tb = t == 0 && dbg ? now() : 0;
      // groups in flight per thread: as many as hide the misses of short groups, few enough that every thread gets seeds
      const int in_flight = (int)std::max<int64_t>(1, std::min<int64_t>(8, n_seeds / (2 * T)));
      // grow the round's groups against the frozen state, several in flight per thread; reserve every member for the
      // earliest group that wants it
      int active = 0;
      bool drained = false;
      auto refill = [&](Grower& w) {
        if (drained) return false;
        const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n_seeds) {
          drained = true;
          return false;
        }
        w.start(g, seeds[(size_t)i], i);
        return true;
      };
      for (int k = 0; k < 8; ++k) {
        gr[(size_t)k].stage = Grower::kIdle;
        if (k < in_flight && refill(gr[(size_t)k])) ++active;
      }
      while (active > 0) {
        for (Grower& w : gr) {
          if (w.stage == Grower::kIdle) continue;
          if (w.advance(g)) continue;
          // (all 16 slots, the unused ones -1: the placement pass counts them, so no per-group size is written here -- neighbouring
          // groups are completed by different threads, and sixteen of their sizes would share a cache line)
          for (int k = w.n; k < 16; ++k) w.out[k] = -1;
          std::memcpy(&cur_members[(size_t)w.index * 16], w.out, sizeof(int32_t) * 16);
          if (n_seeds > 1) {
            const int32_t id = (int32_t)(base + w.index);
            for (int k = 0; k < w.n; ++k) {
              int32_t* c = &claim[(size_t)w.out[k]];
              int32_t cur = load_relaxed(c);
              while (id < cur && !__atomic_compare_exchange_n(c, &cur, id, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
              }
            }
          }
          w.stage = Grower::kIdle;
          if (!refill(w)) --active;
        }
      }
      bar.wait();
      const double tc = t == 0 && dbg ? now() : 0;
      // every group keeps the members it won (order kept) and places them
      const int64_t per = (n_seeds + T - 1) / T;
      for (int64_t i = t * per; i < std::min(n_seeds, (t + 1) * per); ++i) {
        int32_t* out = &cur_members[(size_t)i * 16];
        int n = 0;
        while (n < 16 && out[n] >= 0) ++n;
        int keep = 0;
        for (int k = 0; k < n; ++k) {
          const int32_t v = out[k];
          if (n_seeds > 1 && claim[(size_t)v] != (int32_t)(base + i)) continue;
          out[keep++] = v;
          __atomic_fetch_or(&placed[(uint32_t)v >> 6], (uint64_t)1 << ((uint32_t)v & 63), __ATOMIC_RELAXED);
        }
        sizes[(size_t)(base + i)] = keep;
      }
      bar.wait();
      if (t == 0 && dbg) {
        const double td = now();
        t_sel += tb - ta;
        t_grow += tc - tb;
        t_place += td - tc;
      }
    }
  });

  const double t_groups = now();
  int64_t q = 0, n_groups = 0;
  round_first.push_back((int64_t)sizes.size());
  // groups in ascending seed order: a counting sort by segment, stable in the creation index (= rounds ascending)
  std::vector<int64_t> seg_at((size_t)batch + 1, 0);
  for (int32_t k : group_seg) seg_at[(size_t)k + 1]++;
  for (int64_t k = 0; k < batch; ++k) seg_at[(size_t)k + 1] += seg_at[(size_t)k];
  std::vector<int32_t> by_seed(sizes.size());
  for (size_t i = 0; i < sizes.size(); ++i) by_seed[(size_t)seg_at[(size_t)group_seg[i]]++] = (int32_t)i;
  for (int pass = 0; pass < 2; ++pass)  // full groups first, then the short ones (LOI.cpp:873-891)
    for (int32_t i : by_seed) {
      const int n = sizes[(size_t)i];
      if ((n == 16) != (pass == 0)) continue;
      const size_t r = (size_t)group_round[(size_t)i];
      const int32_t* m = round_members[r].get() + (size_t)(i - round_first[r]) * 16;
      for (int k = 0; k < n; ++k) perm_out[q++] = m[k];
    }
  for (int64_t i = 0; i < N; ++i)
    if (!is_placed(placed.data(), (int32_t)i)) perm_out[q++] = (int32_t)i;
  for (int32_t i : by_seed)
    if (sizes[(size_t)i] > 0) {  // (a group that lost every member to earlier seeds of its round is no group)
      if (group_sizes_out) group_sizes_out[n_groups] = sizes[(size_t)i];
      ++n_groups;
    }
  if (n_groups_out) *n_groups_out = n_groups;
  if (dbg)
    std::fprintf(stderr, "loi_fast: T=%d (%s%s) batch=%lld cap=%d: in-CSR %.3f s, groups %.3f s (seed selection %.3f, growth %.3f, placement %.3f), output %.3f s\n", T,
                 pinned ? "kept on the caller's " : "not pinned", pinned ? pin_mode : "", (long long)batch, list_cap, t_incsr - t_start, t_groups - t_incsr, t_sel, t_grow, t_place, now() - t_groups);
  return q == N ? HCSPMM_OK : HCSPMM_EINVAL;
}
