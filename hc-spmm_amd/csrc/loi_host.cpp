// loi_host.cpp -- host-side LOI layout reorder (hcspmm_loi_reorder, hcspmm_apply_permutation).
//
// Produces, bit for bit, the vertex order that the reference's LOI.cpp main writes to
// reorder_direct.txt when it runs reorder_plus_new_direct (LOI.cpp:660-805, output order
// :873-891, in-CSR :826-841): greedy groups of up to 16 rows, seeded by the first unvisited
// non-empty row, grown 15 times by the candidate (rows sharing a column with the group, in
// discovery order) that maximises (float)(ones + deg v) / (cols + deg v - shared v), first-seen
// winning ties.  Data structures are our own: per-vertex stamps replace the reference's
// per-group N-bit bitmap (LOI.cpp:695, O(N^2/16) overall) and its re-sorted column vector
// (LOI.cpp:71), and there is no static 18 269 000-entry table (LOI.cpp:96).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "hcspmm.h"
#include "host_util.h"
#include "loi_scan.h"

namespace {
int loi_reorder_impl(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant, int32_t* perm_out,
                     int32_t* group_sizes_out, int64_t* n_groups_out);

}

extern "C" int hcspmm_loi_reorder(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int32_t* perm_out,
                                  int32_t* group_sizes_out, int64_t* n_groups_out) {
  return loi_reorder_impl(rowptr, col, N, E, HCSPMM_LOI_NEW_DIRECT, perm_out, group_sizes_out, n_groups_out);
}

namespace {
int loi_reorder_windowed(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, bool direct, int32_t* perm_out,
                         int32_t* group_sizes_out, int64_t* n_groups_out);
}

extern "C" int hcspmm_loi_reorder_variant(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant,
                                          int32_t* perm_out, int32_t* group_sizes_out, int64_t* n_groups_out) {
  if (variant == HCSPMM_LOI_WINDOWED_DIRECT || variant == HCSPMM_LOI_WINDOWED)
    return loi_reorder_windowed(rowptr, col, N, E, variant == HCSPMM_LOI_WINDOWED_DIRECT, perm_out, group_sizes_out, n_groups_out);
  if (variant != HCSPMM_LOI_NEW_DIRECT && variant != HCSPMM_LOI_NEW) return HCSPMM_EINVAL;
  return loi_reorder_impl(rowptr, col, N, E, variant, perm_out, group_sizes_out, n_groups_out);
}

namespace {
int loi_reorder_impl(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant, int32_t* perm_out,
                     int32_t* group_sizes_out, int64_t* n_groups_out) {
  if (N < 0 || E < 0 || !rowptr || (N > 0 && !perm_out) || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  if (!hcspmm::csr_row_pointers_ok(rowptr, N, E)) return HCSPMM_EINVAL;
  for (int64_t e = 0; e < E; ++e)
    if (col[e] < 0 || col[e] >= N) return HCSPMM_EINVAL;

  // rows that reference a column: the in-CSR (in-neighbour lists ascending, row-major scan) for
  // reorder_plus_new_direct; reorder_plus_new (LOI.cpp:505-658) walks the column's OWN out-list
  // instead (LOI.cpp:556-557), i.e. it assumes a symmetric graph
  // The lists are private copies: rows already placed in a group are squeezed out of a list (order kept) the
  // next time it is walked, so a hub column's list shrinks as the run proceeds instead of being re-read in full
  // by every group that touches the hub.
  std::vector<int32_t> rowptr_in_v, col_in_v;
  const int32_t* rowptr_in = rowptr;
  if (variant != HCSPMM_LOI_NEW_DIRECT) {
    col_in_v.assign(col, col + E);
  } else {
    rowptr_in_v.assign((size_t)N + 1, 0);
    col_in_v.resize((size_t)E);
    for (int64_t e = 0; e < E; ++e) rowptr_in_v[(size_t)col[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr_in_v[(size_t)i + 1] += rowptr_in_v[(size_t)i];
    std::vector<int32_t> fill(rowptr_in_v.begin(), rowptr_in_v.end() - 1);
    for (int64_t r = 0; r < N; ++r)
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) col_in_v[(size_t)fill[(size_t)col[e]]++] = (int32_t)r;
    rowptr_in = rowptr_in_v.data();
  }
  int32_t* col_in = col_in_v.data();
  std::vector<int32_t> in_len((size_t)N);
  for (int64_t c = 0; c < N; ++c) in_len[(size_t)c] = rowptr_in[c + 1] - rowptr_in[c];

  // Candidates live in position-indexed arrays (discovery order), so that the pricing scan -- 15 passes over
  // every candidate per group, the bulk of the run time on graphs with hub columns -- reads three dense
  // arrays front to back instead of chasing visit / shared / row_pointers per vertex, and can be priced eight
  // at a time (scan_best_avx2: same IEEE conversions and division, same first-seen-wins order).
  std::vector<uint8_t> visit((size_t)N, 0);
  struct Where { int32_t stamp, pos; };            // group id in which v became a candidate, and its position there
  std::vector<Where> where((size_t)N, Where{-1, 0});
  std::vector<int32_t> col_stamp((size_t)N, -1);  // group id in which column c joined the group
  std::vector<int32_t> cand, cand_deg, cand_shared, cand_alive;  // alive: -1 (all bits) or 0, a lane mask
  std::vector<int32_t> resi, next_resi;
  std::vector<std::vector<int32_t>> groups;
  auto deg = [&](int32_t v) { return rowptr[v + 1] - rowptr[v]; };

  int32_t gid = 0;
  int64_t seed_scan = 0;
  for (;;) {
    while (seed_scan < N && (deg((int32_t)seed_scan) == 0 || visit[(size_t)seed_scan])) ++seed_scan;
    if (seed_scan >= N) break;
    const int32_t seed = (int32_t)seed_scan;
    std::vector<int32_t> grp{seed};
    visit[(size_t)seed] = 1;
    cand.clear();
    cand_deg.clear();
    cand_shared.clear();
    cand_alive.clear();
    int32_t ncols = 0;
    resi.clear();
    for (int32_t e = rowptr[seed]; e < rowptr[seed + 1]; ++e) {
      const int32_t c = col[e];
      if (col_stamp[(size_t)c] != gid) {  // duplicate-free rows: always true, kept for safety
        col_stamp[(size_t)c] = gid;
        ++ncols;
      }
      resi.push_back(c);  // the seed's columns are scanned like any residual
    }
    int32_t ones = deg(seed);
    bool first = true;
    for (int step = 0; step < 15; ++step) {
      for (int32_t c : resi) {
        int32_t* lst = col_in + rowptr_in[(size_t)c];
        const int32_t len = in_len[(size_t)c];
        int32_t keep = 0;
        for (int32_t j = 0; j < len; ++j) {
          const int32_t r = lst[j];
          if (!visit[(size_t)r]) {
            lst[keep++] = r;
            Where& wr = where[(size_t)r];
            if (wr.stamp != gid) {
              wr.stamp = gid;
              wr.pos = (int32_t)cand.size();
              cand.push_back(r);
              cand_deg.push_back(deg(r));
              cand_shared.push_back(0);
              cand_alive.push_back(-1);
            }
            cand_shared[(size_t)wr.pos]++;
          }
        }
        in_len[(size_t)c] = keep;
      }
      // first pick: the reference prices against the seed's own entry count (LOI.cpp:726-727):
      //   (ones + deg v) / (ones + deg v - shared v);
      // later picks against the group's distinct-column count (LOI.cpp:775-776):
      //   (ones + deg v) / (ncols + deg v - shared v)
      const int32_t base = first ? ones : ncols;
      const int64_t n_cand = (int64_t)cand.size();
      const int64_t bp = hcspmm::loi::scan_best(cand_deg.data(), cand_shared.data(), cand_alive.data(), n_cand, ones, base);
      if (bp < 0) break;
      const int32_t best = cand[(size_t)bp];
      cand_alive[(size_t)bp] = 0;
      grp.push_back(best);
      visit[(size_t)best] = 1;
      next_resi.clear();
      for (int32_t e = rowptr[best]; e < rowptr[best + 1]; ++e) {
        const int32_t c = col[e];
        if (col_stamp[(size_t)c] != gid) {
          col_stamp[(size_t)c] = gid;
          ++ncols;
          next_resi.push_back(c);
        }
      }
      resi.swap(next_resi);
      ones += deg(best);
      first = false;
    }
    groups.push_back(std::move(grp));
    ++gid;
  }

  int64_t p = 0;
  for (const auto& g : groups)
    if (g.size() == 16) for (int32_t v : g) perm_out[p++] = v;
  for (const auto& g : groups)
    if (g.size() < 16) for (int32_t v : g) perm_out[p++] = v;
  for (int64_t i = 0; i < N; ++i)
    if (!visit[(size_t)i]) perm_out[p++] = (int32_t)i;
  if (group_sizes_out)
    for (size_t i = 0; i < groups.size(); ++i) group_sizes_out[i] = (int32_t)groups[i].size();
  if (n_groups_out) *n_groups_out = (int64_t)groups.size();
  return p == N ? HCSPMM_OK : HCSPMM_EINVAL;
}
}  // namespace

// ------------------------------------------------------------------------------------------
// The windowed variants reorder_plus_direct / reorder_plus (LOI.cpp:286-484 / :98-284; not called by the reference's
// main): rows are ordered by their smallest column id, a group is seeded by the first row of that order not yet
// placed and grown up to 15 times by the best of the NEXT 300 rows of the order (vertex window VW = 300) instead of
// by the best of all rows sharing a column.  Restated with the reference's exact state handling, including what
// looks accidental there, because the output depends on it:
//  * the row order comes from a NON-stable std::sort on the smallest column id (LOI.cpp:301): rows with equal keys
//    land in the order libstdc++'s introsort leaves them in -- reproduced by sorting the same pairs with the same
//    comparison through the same library (the fixtures were generated with this container's libstdc++);
//  * a seed is looked for only among the first F - 50 rows of the order (LOI.cpp:333);
//  * after a group the shared-column counters are cleared for the VERTEX IDS cur .. cur+299 (LOI.cpp:468-472 indexes
//    the per-vertex table by window position) and for the rows touched in this group; a group whose first pick finds
//    no candidate skips both (LOI.cpp:384-387 `continue`), leaving counters -- and, in the _direct variant, the
//    touched marks -- stale.
// Domain: the reference reads past its row-order array when the graph has rows without entries (LOI.cpp:362 bounds the
// window by the vertex count, not by the number of non-empty rows) and underflows F - 50 below 50 non-empty rows, so
// those inputs -- where its behaviour is undefined -- are refused with HCSPMM_EINVAL, as are rows whose columns are
// not strictly ascending (its residual merge, LOI.cpp:36-58, assumes they are).
// ------------------------------------------------------------------------------------------
namespace {
int loi_reorder_windowed(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, bool direct, int32_t* perm_out,
                         int32_t* group_sizes_out, int64_t* n_groups_out) {
  if (N < 0 || E < 0 || !rowptr || (N > 0 && !perm_out) || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  constexpr int64_t kWindow = 300, kTail = 50;
  if (N < kTail) return HCSPMM_EINVAL;
  if (!hcspmm::csr_row_pointers_ok(rowptr, N, E)) return HCSPMM_EINVAL;  // (the per-row loop below then stays inside col[0, E))
  for (int64_t r = 0; r < N; ++r) {
    if (rowptr[r + 1] <= rowptr[r]) return HCSPMM_EINVAL;  // an empty row: outside the reference's defined domain
    for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      if (col[e] < 0 || col[e] >= N) return HCSPMM_EINVAL;
      if (e > rowptr[r] && col[e] <= col[e - 1]) return HCSPMM_EINVAL;
    }
  }
  // rows that reference a column: in-CSR for the _direct variant, the column vertex's own out-list otherwise
  std::vector<int32_t> rowptr_in_v, col_in_v;
  const int32_t *rowptr_in = rowptr, *col_in = col;
  if (direct) {
    rowptr_in_v.assign((size_t)N + 1, 0);
    col_in_v.resize((size_t)E);
    for (int64_t e = 0; e < E; ++e) rowptr_in_v[(size_t)col[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr_in_v[(size_t)i + 1] += rowptr_in_v[(size_t)i];
    std::vector<int32_t> fill(rowptr_in_v.begin(), rowptr_in_v.end() - 1);
    for (int64_t r = 0; r < N; ++r)
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) col_in_v[(size_t)fill[(size_t)col[e]]++] = (int32_t)r;
    rowptr_in = rowptr_in_v.data();
    col_in = col_in_v.data();
  }
  auto deg = [&](int32_t v) { return rowptr[v + 1] - rowptr[v]; };

  // the row order: (row, smallest column id), library sort on the second member only
  std::vector<std::pair<int, int>> order((size_t)N);
  for (int64_t r = 0; r < N; ++r) order[(size_t)r] = std::pair<int, int>((int)r, col[rowptr[r]]);  // ascending row: first = smallest
  std::sort(order.begin(), order.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });

  std::vector<uint8_t> visit((size_t)N, 0), touched((size_t)N, 0);
  std::vector<int32_t> shared((size_t)N, 0), touched_list, col_stamp((size_t)N, -1), resi, next_resi;
  std::vector<std::vector<int32_t>> groups;
  const int64_t max_groups = (N + 15) / 16;
  int64_t cur = 0;
  for (int64_t z = 0; z < max_groups; ++z) {
    while (cur < N - kTail && visit[(size_t)order[(size_t)cur].first]) ++cur;
    if (cur >= N - kTail) break;
    const int32_t seed = order[(size_t)cur].first;
    const int64_t win_end = std::min<int64_t>(cur + kWindow, N);
    std::vector<int32_t> grp{seed};
    visit[(size_t)seed] = 1;
    touched_list.clear();
    auto bump = [&](int32_t c) {  // every row that references column c and is not placed shares one more column
      for (int32_t j = rowptr_in[c]; j < rowptr_in[c + 1]; ++j) {
        const int32_t r = col_in[j];
        if (visit[(size_t)r]) continue;
        shared[(size_t)r] += 1;
        if (!touched[(size_t)r]) {
          touched[(size_t)r] = 1;
          touched_list.push_back(r);
        }
      }
    };
    int32_t ncols = 0;
    for (int32_t e = rowptr[seed]; e < rowptr[seed + 1]; ++e) {
      col_stamp[(size_t)col[e]] = (int32_t)z;
      ++ncols;
      bump(col[e]);
    }
    int32_t ones = deg(seed);
    // first pick prices against the two rows' entry counts (LOI.cpp:365-372), later ones against the group's
    // distinct columns (LOI.cpp:425-433); strict '>' from 0: the first row of the window order wins ties
    auto pick = [&](bool first) {
      int32_t best = -1;
      float best_profit = 0.0f;
      for (int64_t i = cur; i < win_end; ++i) {
        const int32_t v = order[(size_t)i].first;
        if (visit[(size_t)v]) continue;
        const int32_t o = ones + deg(v);
        const int32_t rows = (first ? o : ncols + deg(v)) - shared[(size_t)v];
        const float profit = (float)o / (float)rows;
        if (profit > best_profit) {
          best = v;
          best_profit = profit;
        }
      }
      return best;
    };
    auto add = [&](int32_t v) {
      grp.push_back(v);
      visit[(size_t)v] = 1;
      next_resi.clear();
      for (int32_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
        if (col_stamp[(size_t)col[e]] != (int32_t)z) {
          col_stamp[(size_t)col[e]] = (int32_t)z;
          ++ncols;
          next_resi.push_back(col[e]);
        }
      }
      resi.swap(next_resi);
      ones += deg(v);
    };
    int32_t v = pick(true);
    if (v < 0) {  // nothing left in the window: the reference keeps the lone seed and skips its clean-up
      // (reorder_plus re-creates its touched marks for every group, LOI.cpp:136; reorder_plus_direct keeps one array)
      if (!direct)
        for (int32_t r : touched_list) touched[(size_t)r] = 0;
      groups.push_back(std::move(grp));
      continue;
    }
    add(v);
    for (int h = 0; h < 14; ++h) {
      for (int32_t c : resi) bump(c);
      v = pick(false);
      if (v < 0) break;
      add(v);
    }
    for (int64_t i = cur; i < win_end; ++i) shared[(size_t)i] = 0;  // by vertex id, as the reference does
    for (int32_t r : touched_list) {
      shared[(size_t)r] = 0;
      touched[(size_t)r] = 0;
    }
    groups.push_back(std::move(grp));
  }
  int64_t p = 0;
  for (const auto& g : groups)
    if (g.size() == 16) for (int32_t v : g) perm_out[p++] = v;
  for (const auto& g : groups)
    if (g.size() < 16) for (int32_t v : g) perm_out[p++] = v;
  for (int64_t i = 0; i < N; ++i)
    if (!visit[(size_t)i]) perm_out[p++] = (int32_t)i;
  if (group_sizes_out)
    for (size_t i = 0; i < groups.size(); ++i) group_sizes_out[i] = (int32_t)groups[i].size();
  if (n_groups_out) *n_groups_out = (int64_t)groups.size();
  return p == N ? HCSPMM_OK : HCSPMM_EINVAL;
}
}  // namespace

extern "C" int hcspmm_apply_permutation(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E,
                                        const int32_t* perm, int32_t* rowptr_out, int32_t* col_out) {
  if (N < 0 || E < 0 || !rowptr || !rowptr_out || (N > 0 && !perm) || (E > 0 && (!col || !col_out)))
    return HCSPMM_EINVAL;
  if (!hcspmm::csr_row_pointers_ok(rowptr, N, E)) return HCSPMM_EINVAL;
  // every pass below is a loop over independent vertices or entries, cut into contiguous ranges for the host threads (on the
  // RD-sized graph the sequential checks and the inverse permutation were most of the call)
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16, hcspmm::host_threads()), E / 65536));
  auto parallel = [&](int64_t n, auto&& body) {  // body(begin, end)
    if (T == 1 || n < 4096) {
      body((int64_t)0, n);
      return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { body(n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
  };
  std::atomic<int> bad{0};
  parallel(E, [&](int64_t a, int64_t b) {
    for (int64_t e = a; e < b; ++e)
      if (col[e] < 0 || col[e] >= N) bad.store(1, std::memory_order_relaxed);
  });
  if (bad.load()) return HCSPMM_EINVAL;
  std::vector<int32_t> inv((size_t)N);
  parallel(N, [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) inv[(size_t)i] = -1;
  });
  parallel(N, [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {
      const int32_t old = perm[i];
      if (old < 0 || old >= N) {
        bad.store(1, std::memory_order_relaxed);
        continue;
      }
      __atomic_store_n(&inv[(size_t)old], (int32_t)i, __ATOMIC_RELAXED);
    }
  });
  // a permutation iff every old id was named, i.e. iff nothing is left at -1 (an id named twice leaves another one unnamed)
  parallel(N, [&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {
      if (inv[(size_t)i] < 0) bad.store(1, std::memory_order_relaxed);
      if (!bad.load(std::memory_order_relaxed)) rowptr_out[i + 1] = rowptr[perm[i] + 1] - rowptr[perm[i]];  // (degrees first, summed below)
    }
  });
  if (bad.load()) return HCSPMM_EINVAL;
  rowptr_out[0] = 0;
  for (int64_t i = 0; i < N; ++i) rowptr_out[i + 1] += rowptr_out[i];
  // new rows, in contiguous ranges balanced by entries
  auto fill_rows = [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i) {
      const int32_t old = perm[i];
      int32_t o = rowptr_out[i];
      for (int32_t e = rowptr[old]; e < rowptr[old + 1]; ++e) col_out[o++] = inv[(size_t)col[e]];
      if (o - rowptr_out[i] > 1) std::sort(col_out + rowptr_out[i], col_out + o);
    }
  };
  if (T == 1) {
    fill_rows(0, N);
  } else {
    std::vector<std::thread> th;
    int64_t i0 = 0;
    for (int t = 0; t < T; ++t) {
      const int64_t i1 = t == T - 1 ? N : std::upper_bound(rowptr_out, rowptr_out + N + 1, (int32_t)(E * (t + 1) / T)) - rowptr_out - 1;
      th.emplace_back(fill_rows, i0, std::max(i0, i1));
      i0 = std::max(i0, i1);
    }
    for (auto& x : th) x.join();
  }
  return HCSPMM_OK;
}
