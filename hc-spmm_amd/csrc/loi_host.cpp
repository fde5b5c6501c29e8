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
#include <cstdint>
#include <vector>

#include "hcspmm.h"

namespace {
int loi_reorder_impl(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant, int32_t* perm_out,
                     int32_t* group_sizes_out, int64_t* n_groups_out);
}

extern "C" int hcspmm_loi_reorder(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int32_t* perm_out,
                                  int32_t* group_sizes_out, int64_t* n_groups_out) {
  return loi_reorder_impl(rowptr, col, N, E, HCSPMM_LOI_NEW_DIRECT, perm_out, group_sizes_out, n_groups_out);
}

extern "C" int hcspmm_loi_reorder_variant(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant,
                                          int32_t* perm_out, int32_t* group_sizes_out, int64_t* n_groups_out) {
  if (variant != HCSPMM_LOI_NEW_DIRECT && variant != HCSPMM_LOI_NEW) return HCSPMM_EINVAL;
  return loi_reorder_impl(rowptr, col, N, E, variant, perm_out, group_sizes_out, n_groups_out);
}

namespace {
int loi_reorder_impl(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int variant, int32_t* perm_out,
                     int32_t* group_sizes_out, int64_t* n_groups_out) {
  if (N < 0 || E < 0 || !rowptr || (N > 0 && !perm_out) || (E > 0 && !col)) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  for (int64_t e = 0; e < E; ++e)
    if (col[e] < 0 || col[e] >= N) return HCSPMM_EINVAL;

  // rows that reference a column: the in-CSR (in-neighbour lists ascending, row-major scan) for
  // reorder_plus_new_direct; reorder_plus_new (LOI.cpp:505-658) walks the column's OWN out-list
  // instead (LOI.cpp:556-557), i.e. it assumes a symmetric graph
  std::vector<int32_t> rowptr_in_v, col_in_v;
  const int32_t* rowptr_in = rowptr;
  const int32_t* col_in = col;
  if (variant == HCSPMM_LOI_NEW_DIRECT) {
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

  std::vector<uint8_t> visit((size_t)N, 0);
  std::vector<int32_t> shared((size_t)N, 0);      // the reference's `cns`
  std::vector<int32_t> cand_stamp((size_t)N, -1); // group id in which v became a candidate
  std::vector<int32_t> col_stamp((size_t)N, -1);  // group id in which column c joined the group
  std::vector<int32_t> cand, resi, next_resi;
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
        for (int32_t j = rowptr_in[(size_t)c]; j < rowptr_in[(size_t)c + 1]; ++j) {
          const int32_t r = col_in[(size_t)j];
          if (!visit[(size_t)r]) {
            shared[(size_t)r]++;
            if (cand_stamp[(size_t)r] != gid) {
              cand_stamp[(size_t)r] = gid;
              cand.push_back(r);
            }
          }
        }
      }
      int32_t best = -1;
      float best_profit = 0.0f;
      for (int32_t v : cand) {
        if (visit[(size_t)v]) continue;
        const int32_t o = ones + deg(v);
        // first pick: the reference prices against the seed's own entry count (LOI.cpp:726-727),
        // later picks against the group's distinct-column count (LOI.cpp:775-776)
        const int32_t rws = first ? (o - shared[(size_t)v]) : (ncols + deg(v) - shared[(size_t)v]);
        const float profit = (float)o / (float)rws;
        if (profit > best_profit) {
          best = v;
          best_profit = profit;
        }
      }
      if (best < 0) break;
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
    for (int32_t v : cand) shared[(size_t)v] = 0;
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

extern "C" int hcspmm_apply_permutation(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E,
                                        const int32_t* perm, int32_t* rowptr_out, int32_t* col_out) {
  if (N < 0 || E < 0 || !rowptr || !rowptr_out || (N > 0 && !perm) || (E > 0 && (!col || !col_out)))
    return HCSPMM_EINVAL;
  std::vector<int32_t> inv((size_t)N, -1);
  for (int64_t i = 0; i < N; ++i) {
    const int32_t old = perm[i];
    if (old < 0 || old >= N || inv[(size_t)old] != -1) return HCSPMM_EINVAL;
    inv[(size_t)old] = (int32_t)i;
  }
  rowptr_out[0] = 0;
  for (int64_t i = 0; i < N; ++i) {
    const int32_t old = perm[i];
    int32_t o = rowptr_out[i];
    for (int32_t e = rowptr[old]; e < rowptr[old + 1]; ++e) col_out[o++] = inv[(size_t)col[e]];
    std::sort(col_out + rowptr_out[i], col_out + o);
    rowptr_out[i + 1] = o;
  }
  return HCSPMM_OK;
}
