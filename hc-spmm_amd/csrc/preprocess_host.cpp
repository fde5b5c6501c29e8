// preprocess_host.cpp -- host-side window condensing + classifier (hcspmm_preprocess_host).
//
// Produces the four integer products of the reference's `preprocess`
// (hybrid_kernel/hybrid_all_kernel.cu:339-408): edgeToRow (:314-326), and per 16-row window the
// number of 8-column blocks of its condensed form, the sparse/dense class and each entry's
// condensed column (:242-269, on the per-window sorted copy that :386-399 makes with
// thrust::sort).  The reference does this with one GPU thread per window; north_star puts it on
// the host, so here windows are spread over host threads in nnz-balanced contiguous ranges, and
// per window the sorted unique column list is built once and every row is ranked against it with
// a linear merge (rows are ascending, so no per-entry binary search is needed).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "hcspmm.h"
#include "host_util.h"

namespace {

// The classifier expression of hybrid_all_kernel.cu:261-262 with the reference's C types:
// (float)size * <double> - ((float)nnz / (int)(num*16*8)) * <double> - <double>; the quotient is
// a float division.  Built with -ffp-contract=off so no term is fused.
inline int classify(int32_t size, uint32_t nnz, int32_t num, int rule) {
  const double t1 = (double)((float)size) * 0.19854024;
  const float dens = (float)nnz / (float)(num * HCSPMM_BLK_H * HCSPMM_BLK_W);
  const double t2 = (double)dens * 6.578043;
  const double logit = (t1 - t2) - 3.14922857;
  if (rule == HCSPMM_RULE_MI355X || rule == HCSPMM_RULE_MI355X_WIDE) {
    // refit on MI355X against this library's two sub-paths (tools/refit_classifier.py, 65536 windows,
    // profiles/r01/classifier_refit_v3.json):  z = w1*size + w2*density + b, sparse-row path when z > 0.
    // The boundary depends on the embedding width (at D = 32 the sparse-row path wins below 10-20 % tile
    // density, at D = 128 the dense-tile path wins almost everywhere) and on whether the window gets a compact
    // record (at most 40 padded columns: two round trips per unit instead of three), so there is one
    // coefficient set per (width class, record kind).  Three sets use the density alone: their free fits had a
    // slightly negative w1 (noise inside K <= 130) that extrapolated hub windows onto the dense-tile path.
    // Two sets were then corrected against whole graphs (profiles/r01/classifier_rules_on_workloads.log): the
    // synthetic grid (random columns) undervalues the dense-tile path on graphs with local structure, so the
    // narrow/compact boundary sits at 7 % density (between the grid's two lowest densities) instead of the
    // fitted 10 %, and the wide/regular set is the v2 fit, which the TT-sized graph prefers to the v3 one.
    const bool wide = rule == HCSPMM_RULE_MI355X_WIDE;
    const bool compact = num * HCSPMM_BLK_W <= HCSPMM_COMPACT_K;
    double w1 = 0.0, w2, b;
    if (!wide && compact) { w2 = -30.904771063924702; b = 2.163333974474729; }
    else if (!wide) { w2 = -91.39571130769644; b = 18.406719120321213; }
    else if (compact) { w2 = -39.36167079949795; b = -0.21260854261350828; }
    else { w1 = 0.030703533058157952; w2 = -139.72170588602881; b = 4.259271957775277; }
    const double z = ((double)((float)size) * w1 + (double)dens * w2) + b;
    return z > 0 ? 0 : 1;
  }
  switch (rule) {
    case HCSPMM_RULE_INTENDED: return logit > 0 ? 0 : 1;
    case HCSPMM_RULE_INTENDED_GUARD: return (size > 32 || logit > 0) ? 0 : 1;
    default: return logit != 0.0 ? 0 : 1;
  }
}

// *bad is set when a column id lies outside [0, M): the caller turns it into HCSPMM_EINVAL (the pass already
// touches every entry, so the check is free; on the GPU the same id would be an out-of-bounds gather).
// The row pointers are checked here too, window by window and before a single column id of the window is read: inside [0, E] and
// never decreasing (the caller has checked rowptr[0] == 0 and rowptr[N] == E) -- as a sequential loop ahead of the threads this was
// 2 ms of the RD-sized graph's 7.
struct Scratch {  // a thread's buffers, kept across the chunks of windows it takes
  std::vector<int32_t> uniq, buf_a, buf_b;
};
void process_windows(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int64_t M, int64_t w_begin, int64_t w_end,
                     int rule, int32_t* blockPartition, int32_t* edgeToColumn, int32_t* edgeToRow, int32_t* hybrid_type,
                     int* bad, Scratch& scratch) {
  std::vector<int32_t>&uniq = scratch.uniq, &buf_a = scratch.buf_a, &buf_b = scratch.buf_b;
  for (int64_t w = w_begin; w < w_end; ++w) {
    const int64_t r0 = w * HCSPMM_BLK_H, r1 = std::min<int64_t>(r0 + HCSPMM_BLK_H, N);
    const int64_t lo = rowptr[r0], hi = rowptr[r1];
    {
      bool ok = lo >= 0 && hi <= E;
      for (int64_t r = r0; r < r1; ++r) ok &= rowptr[r + 1] >= rowptr[r];
      if (!ok) { *bad = 1; return; }
    }
    {
      uint32_t over = 0;  // unsigned compare: negative ids are "over" too
      for (int64_t e = lo; e < hi; ++e) over |= (uint32_t)((uint32_t)col[e] >= (uint32_t)M);
      if (over) { *bad = 1; return; }
    }
    if (edgeToRow)  // optional: callers with the graph in HBM expand row ids there instead
      for (int64_t r = r0; r < r1; ++r)
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) edgeToRow[e] = (int32_t)r;
    if (hi == lo) {  // reference returns early and leaves garbage (K.cu:252-253); defined as 0/0
      blockPartition[w] = 0;
      hybrid_type[w] = 0;
      continue;
    }
    // The window's ascending unique column list (what the reference gets from thrust::sort + in-place dedupe,
    // K.cu:386-399, :213-223).  Rows arrive ascending (dataset.py's scipy tocsr), so the list is a 16-way merge:
    // a tree of pairwise merges over two scratch buffers, four passes over the window's entries instead of a sort
    // (which was most of the host pass: 3x on the Reddit-scale graph).  A row that is not ascending falls back to
    // the sort, so any input gives the same list.
    bool rows_ascending = true;
    for (int64_t r = r0; r < r1 && rows_ascending; ++r)
      for (int64_t e = (int64_t)rowptr[r] + 1; e < rowptr[r + 1]; ++e)
        if (col[e - 1] > col[e]) { rows_ascending = false; break; }
    if (!rows_ascending) {
      uniq.assign(col + lo, col + hi);
      std::sort(uniq.begin(), uniq.end());
    } else {
      const size_t m = (size_t)(hi - lo);
      buf_a.assign(col + lo, col + hi);
      buf_b.resize(m);
      int64_t cuts[HCSPMM_BLK_H + 1];  // run boundaries, relative to lo
      int runs = 0;
      for (int64_t r = r0; r <= r1; ++r) cuts[runs++] = (int64_t)rowptr[r] - lo;
      --runs;  // number of rows = number of sorted runs
      int32_t *src = buf_a.data(), *dst = buf_b.data();
      while (runs > 1) {
        int out_runs = 0;
        for (int i = 0; i + 1 < runs; i += 2) {
          std::merge(src + cuts[i], src + cuts[i + 1], src + cuts[i + 1], src + cuts[i + 2], dst + cuts[i]);
          cuts[out_runs++] = cuts[i];
        }
        if (runs & 1) {
          std::copy(src + cuts[runs - 1], src + cuts[runs], dst + cuts[runs - 1]);
          cuts[out_runs++] = cuts[runs - 1];
        }
        cuts[out_runs] = (int64_t)m;
        runs = out_runs;
        std::swap(src, dst);
      }
      uniq.resize(m);
      uniq.erase(std::unique_copy(src, src + m, uniq.begin()), uniq.end());  // (de-duplicated on the way out of the merge buffer)
    }
    if (!rows_ascending) uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int32_t size = (int32_t)uniq.size() - 1;                      // K.cu:256-257
    const int32_t num = (size + HCSPMM_BLK_W) / HCSPMM_BLK_W;           // K.cu:258
    blockPartition[w] = num;
    hybrid_type[w] = classify(size, (uint32_t)(hi - lo), num, rule);
    for (int64_t r = r0; r < r1; ++r) {
      const int64_t a = rowptr[r], b = rowptr[r + 1];
      bool ascending = rows_ascending;  // (checked once for the whole window above)
      if (!ascending) {
        ascending = true;
        for (int64_t e = a + 1; e < b; ++e) ascending &= col[e - 1] <= col[e];
      }
      if (ascending) {  // merge walk: both sequences ascending
        size_t p = 0;
        for (int64_t e = a; e < b; ++e) {
          while (uniq[p] < col[e]) ++p;
          edgeToColumn[e] = (int32_t)p;
        }
      } else {  // unsorted row (not produced by dataset.py, but the reference's search copes)
        for (int64_t e = a; e < b; ++e)
          edgeToColumn[e] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), col[e]) - uniq.begin());
      }
    }
  }
}

}  // namespace

extern "C" int hcspmm_preprocess_host(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int64_t M, int rule,
                                      int num_threads, int32_t* blockPartition, int32_t* edgeToColumn,
                                      int32_t* edgeToRow, int32_t* hybrid_type) {
  if (N < 0 || E < 0 || !rowptr) return HCSPMM_EINVAL;
  if (M <= 0) M = N;  // the reference's graphs are square
  if (M > INT32_MAX) return HCSPMM_ERANGE;
  if (E > 0 && (!col || !edgeToColumn)) return HCSPMM_EINVAL;  // edgeToRow may be NULL (skipped)
  if (rule < HCSPMM_RULE_INTENDED || rule > HCSPMM_RULE_MI355X_WIDE) return HCSPMM_EINVAL;
  if (N > INT32_MAX - 16 || E > INT32_MAX) return HCSPMM_ERANGE;
  const int64_t W = (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H;
  if (W > 0 && (!blockPartition || !hybrid_type)) return HCSPMM_EINVAL;
  if (rowptr[0] != 0 || rowptr[N] != E) return HCSPMM_EINVAL;  // (monotonicity: per window, in process_windows)

  int T = num_threads > 0 ? num_threads : hcspmm::host_threads();
  if (T < 1) T = 1;
  if (W < 4 * T || E < (1 << 16)) T = 1;
  if (T == 1) {
    int bad = 0;
    Scratch scratch;
    process_windows(rowptr, col, N, E, M, 0, W, rule, blockPartition, edgeToColumn, edgeToRow, hybrid_type, &bad, scratch);
    return bad ? HCSPMM_EINVAL : HCSPMM_OK;
  }
  // Chunks of consecutive windows with about equal (entries + windows) each, many more than threads, handed out by an atomic counter:
  // the cost of a window is not linear in its entries (a hub window's merge passes miss the L1), and fixed ranges left the launch
  // waiting for its slowest thread (12 ms on 16 threads where 64 took 4.5).  Every output is per window: the result does not depend
  // on who takes which chunk.
  const int n_chunks = (int)std::min<int64_t>(W, (int64_t)T * 32);
  std::vector<int64_t> cut((size_t)n_chunks + 1, W);
  cut[0] = 0;
  const double total = (double)E + (double)W;
  int64_t w = 0;
  for (int c = 1; c < n_chunks; ++c) {
    const double target = total * c / n_chunks;
    while (w < W && (double)rowptr[std::min<int64_t>(w * HCSPMM_BLK_H, N)] + (double)w < target) ++w;
    cut[(size_t)c] = w;
  }
  std::vector<std::thread> th;
  std::vector<int> bad((size_t)T, 0);
  std::atomic<int> next{0};
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      Scratch scratch;
      for (;;) {
        const int c = next.fetch_add(1, std::memory_order_relaxed);
        if (c >= n_chunks || bad[(size_t)t]) break;
        process_windows(rowptr, col, N, E, M, cut[(size_t)c], cut[(size_t)c + 1], rule, blockPartition, edgeToColumn, edgeToRow, hybrid_type,
                        &bad[(size_t)t], scratch);
      }
    });
  for (auto& x : th) x.join();
  for (int t = 0; t < T; ++t)
    if (bad[(size_t)t]) return HCSPMM_EINVAL;
  return HCSPMM_OK;
}
