/*
 * hcspmm.h -- C ABI of libhcspmm.so: MI355X-native (gfx950) hybrid SpMM for GNN
 * aggregation, the drop-in for the hot path of ZJU-DAILY/HC-SpMM.
 *
 * Every entry point names the reference interface it replaces (file:line under
 * the reference tree; "K.cu" = hybrid_kernel/hybrid_all_kernel.cu, "B.cpp" =
 * hybrid_kernel/hybrid_all.cpp).  The ABI is plain C: raw pointers, sizes and a
 * HIP stream; no torch types, no exceptions, no global mutable state (this describes the C ABI: the two Python-visible
 * front-ends above it each keep a process-wide, weakly-referencing registry of the plans they made).  All
 * functions return HCSPMM_OK (0) or a negative code; hcspmm_strerror() names it.
 *
 * Pointer naming: *_h = host memory, *_d = device (HBM) memory.
 * Index arrays are int32 (reference: dataset.py:102-103, K.cu:439-443); features
 * are fp32 row-major (hcspmm_forward_typed also takes fp16 / bf16).  A is binary: edge
 * values are never read.
 */
#ifndef HCSPMM_H
#define HCSPMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 = the signatures below.  Entry points ADDED since 3 was introduced leave it unchanged (a consumer built against an older header keeps
 * working): hcspmm_loi_reorder_fast, hcspmm_dense_update (round 4).  HCSPMM_RULE_MI355X as the front-ends' default classifier is a front-end
 * matter: every C entry point that classifies takes its rule as an argument. */
#define HCSPMM_ABI_VERSION 3

/* Row-window geometry: hybrid_kernel/config.h:4-5 (BLK_H 16, BLK_W 8). */
#define HCSPMM_BLK_H 16
#define HCSPMM_BLK_W 8

/* return codes */
#define HCSPMM_OK 0
#define HCSPMM_EINVAL (-1)     /* bad argument (null pointer, negative size, D <= 0, a column id outside [0, num_columns), ...) */
#define HCSPMM_ENOMEM (-2)     /* host allocation failed */
#define HCSPMM_EPLAN (-3)      /* plan blob does not match this graph (magic, version, N, E, section bounds, fingerprint) */
#define HCSPMM_EHIP (-4)       /* a HIP runtime call or kernel launch failed (hcspmm_last_hip_error) */
#define HCSPMM_EWORKSPACE (-5) /* workspace smaller than hcspmm_workspace_bytes() */
#define HCSPMM_ERANGE (-6)     /* a size exceeds what the int32 index contract can address */

/* Window classifier rule. */
#define HCSPMM_RULE_INTENDED 0       /* logit > 0 -> sparse-row(0) else dense-tile(1); paper p.7, K.cu:261 w/o guard */
#define HCSPMM_RULE_INTENDED_GUARD 1 /* K.cu:261 literally: size > 32 || logit > 0 -> 0 */
#define HCSPMM_RULE_AS_SHIPPED 2     /* K.cu:262 literally: float used as bool */
#define HCSPMM_RULE_MI355X 3         /* same two features, coefficients refit on MI355X with the paper's procedure
                                        against THIS library's two sub-paths (tools/refit_classifier.py,
                                        profiles/r01/classifier_refit_v3.json) at embedding width 32: for
                                        narrow embeddings (D < 64); not a reference output */
#define HCSPMM_RULE_MI355X_WIDE 4    /* the same refit at embedding width 128: for D >= 64 (preprocess does not
                                        know D -- the reference's signature has none -- so the caller picks) */

const char* hcspmm_strerror(int code);
int hcspmm_abi_version(void);
/* hipError_t (as int) of the most recent failing HIP call on this thread, 0 if none. */
int hcspmm_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * preprocess (host).  Replaces `preprocess` K.cu:339-408 (binding B.cpp:13-17, :501) and its
 * kernels fill_edgeToRow K.cu:314-326, fill_segment K.cu:289-301, thrust::sort K.cu:386-399,
 * generate_edgetocolumn K.cu:242-269.  Integer outputs are bit-exact with the reference
 * algorithm (Appendix A of SURVEY.md); empty windows get blockPartition = hybrid_type = 0.
 *   row_pointers_h[N+1], column_index_h[E] : CSR, columns ascending & unique within a row
 *   num_columns : rows of the matrix the column ids index (the reference's graphs are square: pass num_nodes, or
 *                 <= 0 for the same); a multi-GPU row block passes the height of the gathered embedding matrix.
 *                 Every column id is checked against [0, num_columns) -- HCSPMM_EINVAL otherwise -- so that a
 *                 malformed CSR is an error here instead of an out-of-bounds gather on the GPU (the reference
 *                 checks nothing).
 *   blockPartition_h[W], hybrid_type_h[W], edgeToColumn_h[E], edgeToRow_h[E] : outputs, W = ceil(N/16)
 *   num_threads <= 0 : HCSPMM_THREADS if set, else min(64, host hardware threads); the outputs do not depend on it.
 *   edgeToRow_h may be NULL (not produced): it is the plain CSR row expansion, which a caller whose
 *   graph lives in HBM can generate there without a host round trip.
 * ---------------------------------------------------------------------------------------- */
int hcspmm_preprocess_host(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                           int64_t num_edges, int64_t num_columns, int rule, int num_threads,
                           int32_t* blockPartition_h, int32_t* edgeToColumn_h, int32_t* edgeToRow_h,
                           int32_t* hybrid_type_h);

/* edgeToRow on the device: edgeToRow_d[e] = the row that owns stored entry e (the plain CSR row expansion).  Replaces
 * fill_edgeToRow K.cu:314-337 for callers whose graph already lives in HBM (hcspmm_preprocess_host then gets
 * edgeToRow_h = NULL): one thread per entry, a binary search in row_pointers (which sits in L2).  Rows without
 * entries are skipped correctly.  Asynchronous on `stream`. */
int hcspmm_edge_to_row_device(const int32_t* row_pointers_d, int64_t num_nodes, int64_t num_edges, int32_t* edgeToRow_d,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch plan (host build, device resident).  New on MI355X: the reference branches per thread
 * block on hybrid_type (K.cu:960,1039) with one block per window; here the host turns the
 * classified windows into (a) sparse-row tasks ordered by power-of-two length class (row order inside a
 * class), long rows split into segments, and
 * (b) packed dense-tile windows (sorted unique columns + 0/1 tile masks in MFMA lane order).
 * The blob travels in the reference's reserved `row_nzr` tensor slot (K.cu:405: row_nzr/col_nzr
 * are [0] placeholders that every forward receives and never reads).
 *
 * The first HCSPMM_PLAN_HEADER_WORDS int32 words of the blob are a hcspmm_plan_header; callers
 * keep a host copy of it (hcspmm_forward needs the counts to size its grid without a device
 * read-back).
 * ---------------------------------------------------------------------------------------- */
#define HCSPMM_PLAN_MAGIC 0x48435350 /* "HCSP" */
#define HCSPMM_PLAN_VERSION 7
#define HCSPMM_TINY_LEN 2 /* tasks of at most this many entries carry their indices in the descriptor */
#define HCSPMM_COMPACT_K 40     /* dense windows of at most this many (padded) columns use compact records ... */
#define HCSPMM_COMPACT_WORDS 64 /* ... of this many words: [window, K/4, U[40], 10 x (mask lo, mask hi), pad] */
#define HCSPMM_COMPACT2_K 80    /* wider windows up to this many (padded) columns use double records ... */
#define HCSPMM_COMPACT2_WORDS 128 /* ... of this many words: [window, K/4, U[80], 20 x (mask lo, mask hi), pad] */
#define HCSPMM_PLAN_HEADER_WORDS 64

typedef struct hcspmm_plan_header {
  int32_t magic;            /* HCSPMM_PLAN_MAGIC */
  int32_t version;          /* HCSPMM_PLAN_VERSION */
  int32_t total_words;      /* size of the whole blob in int32 words */
  int32_t num_nodes;        /* N the plan was built for */
  int32_t num_edges;        /* E the plan was built for */
  int32_t num_windows;      /* W */
  int32_t split_threshold;  /* rows with more entries than this are split ... */
  int32_t segment_len;      /* ... into segments of this many entries */
  int32_t n_tasks;          /* sparse-row tasks (one per whole row or row segment), by descending length class */
  int32_t n_dense;          /* dense-tile windows */
  int32_t n_split_rows;     /* rows whose partial sums are combined by the fix-up pass */
  int32_t n_partials;       /* partial-sum slots (rows of the workspace) */
  int32_t off_tasks;        /* word offsets of the sections inside the blob */
  int32_t off_dense_index;
  int32_t off_dense_pack;
  int32_t off_fixups;
  int32_t nnz_sparse;       /* entries handled by the sparse-row path */
  int32_t nnz_dense;        /* entries handled by the dense-tile path */
  int32_t uniq_dense;       /* sum of unique columns over dense windows (gathered X rows) */
  int32_t max_dense_k;      /* largest padded K (8*blockPartition) among dense windows */
  int32_t n_len_gt[5];      /* tasks longer than 16, 32, 64, 128, 256 entries (class boundaries are powers
                               of two, so these are prefix sizes of the task list: the launch can hand the n longest tasks to
                               whole waves -- "wide" tasks -- at any of these thresholds) */
  int32_t n_tiny;           /* the last n_tiny tasks have at most HCSPMM_TINY_LEN entries and carry their column indices
                               inline: (row, or -(slot+1) for a row segment | index0 | length | index1), absent = -1 */
  int32_t n_dense_compact;  /* the last n_dense_compact dense windows have K <= HCSPMM_COMPACT_K and live in fixed
                               HCSPMM_COMPACT_WORDS-word records: [window, K/4, U[40], 10 x (mask lo, mask hi), pad] */
  int32_t off_dense_compact; /* word offset of those records (a multiple of 64) */
  int32_t n_dense_compact2; /* the n_dense_compact2 dense windows before them have HCSPMM_COMPACT_K < K <= HCSPMM_COMPACT2_K
                               and live in HCSPMM_COMPACT2_WORDS-word records of the same layout (U[80], 20 masks) */
  int32_t off_dense_compact2; /* word offset of those records (a multiple of 64) */
  int32_t num_columns;      /* rows of X this plan gathers from: every column id it holds or refers to is < this */
  int32_t n_sparse_windows; /* row windows NOT on the dense-tile path (sparse-row and empty ones), ... */
  int32_t off_sparse_windows; /* ... their ids, ascending: the 16-row tiles the update pass of
                               hcspmm_forward_fused still has to multiply by the weights */
  uint32_t fingerprint_lo;  /* hcspmm_graph_fingerprint_host(row_pointers, column_index) of the graph the plan was */
  uint32_t fingerprint_hi;  /* built from: a plan for another graph with the same N and E is told apart by it */
  int32_t dense_k_sum;      /* sum over dense windows of K = 8*blockPartition (padded condensed columns): the
                               dense-tile path executes exactly 2*16*K*D flop per window */
  int32_t flags;            /* HCSPMM_PLAN_FUSE_IN_LAUNCH: the fused operators update this plan's dense-tile windows inside the
                               hybrid launch (hcspmm_plan_params.fuse_in_launch) */
  /* XCD-affine column slices: sparse-path rows longer than slice_threshold are cut where their
   * (ascending) column ids cross n_slices - 1 boundaries and every segment_len entries; the pieces of slice s form
   * their own task list, served only by workgroups with blockIdx % 8 == s % 8 -- workgroups that share one of the
   * eight per-XCD L2s -- so that L2 holds 1/8 of the X rows those tasks gather.  Placement is a speed matter only. */
  int32_t n_slices;         /* 0: none; else a multiple of 8 */
  int32_t slice_threshold;  /* rows with more entries than this are sliced (when n_slices > 0) */
  int32_t off_slice_table;  /* n_slices + 1 task offsets relative to off_slice_tasks, each a multiple of 64: slice s
                               owns descriptors [table[s], table[s+1]), longest length class first, padded with
                               (-1, 0, 0, -1) */
  int32_t off_slice_tasks;  /* descriptors (row, first entry, length, partial slot or -1), as the ordinary tasks */
  int32_t n_slice_tasks;    /* = table[n_slices], padding included */
  int32_t slice_xcd_tasks;  /* max over x in [0, 8) of the descriptors of the slices s = x (mod 8): sizes the region */
  int32_t nnz_sliced;       /* entries covered by sliced tasks (part of nnz_sparse) */
  int32_t n_sliced_rows;
  int32_t panel_cols;       /* hcspmm_plan_params.panel_cols: 0 = the launch chooses, > 0: feature columns per
                               sparse pass for fp32 features (16-bit features: twice as many), < 0: one pass over all columns */
  int32_t reserved[18];
} hcspmm_plan_header;

#define HCSPMM_PLAN_FUSE_IN_LAUNCH 1
#define HCSPMM_PLAN_FUSE_ROWS 2 /* ... and the sparse-row path as well: the row-tile form (hcspmm_plan_params.fuse_in_launch = 2) */
#define HCSPMM_PLAN_FUSE_NEVER 4 /* always two launches (fuse_in_launch < 0); no flag = the operator chooses per call */

/* Tunables for the plan; zero-initialise for defaults. */
typedef struct hcspmm_plan_params {
  int32_t split_threshold; /* default 512 */
  int32_t segment_len;     /* default 256 */
  int32_t fuse_in_launch;  /* form of hcspmm_forward_fused for this plan (see there): 0 = chosen per call (row tiles when out2 --
                              N x D fp32 -- is 80 MB or more at D <= 64, or 256 MB or more at D <= 128, H <= 32 on a graph
                              with a quarter of its rows in dense-tile windows; else two launches), < 0 = always two launches, 1 = dense-tile
                              windows multiply by the weights inside the hybrid launch (slower on MI355X:
                              profiles/r02/ab_fused.log), 2 = row tiles wherever the shape allows */
  int32_t slice_threshold; /* XCD-affine column slices: 0 = automatic (on for num_columns >= 65536 when at least 5 % of
                              the sparse-path entries sit in rows longer than 256 entries and -- below 250 000 columns --
                              the sparse path holds at least 3 M entries; HCSPMM_SLICE_THRESHOLD in the environment
                              overrides), > 0: rows longer than this are sliced, < 0: off */
  int32_t n_slices;        /* 0 = 8 (HCSPMM_SLICES overrides); rounded up to a multiple of 8, at most 64 */
  int32_t panel_cols;      /* feature columns per pass of the sparse-row path: 0 = chosen at launch from the plan's counts and
                              the embedding width (32 fp32 columns = one cache line per gathered row for wide embeddings, 64
                              when X exceeds the Infinity Cache, one pass for short-row graphs), > 0: this many (a multiple
                              of 16; fp32 -- 16-bit features take twice as many), < 0: one pass.  The best value depends on
                              the graph's size and degree mix in no monotone way (profiles/r03/ab_panel_midsize.log): a
                              caller that will run many steps can measure it (hcspmm.tune_plan in the Python front-end) */
} hcspmm_plan_params;

/* Number of int32 words a plan for this graph needs (so the caller can allocate the tensor). */
int hcspmm_plan_words(const int32_t* row_pointers_h, int64_t num_nodes, int64_t num_edges,
                      const int32_t* blockPartition_h, const int32_t* hybrid_type_h,
                      const hcspmm_plan_params* params, int64_t* words_out);

/* Fill plan_h[words] (host).  The caller uploads it to HBM unchanged.  num_columns as for
 * hcspmm_preprocess_host (<= 0: num_nodes); column ids are range-checked here too (HCSPMM_EINVAL), since a plan
 * may be built for a classification that did not come from hcspmm_preprocess_host. */
int hcspmm_plan_build(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                      int64_t num_edges, int64_t num_columns, const int32_t* blockPartition_h,
                      const int32_t* edgeToColumn_h, const int32_t* hybrid_type_h, const hcspmm_plan_params* params,
                      int32_t* plan_h, int64_t words);

/* Validate a header against (N, E) and its own section layout (every section inside total_words, counts
 * consistent); plan_words_available = length of the buffer that holds the blob (<= 0: not checked).
 * HCSPMM_OK or HCSPMM_EPLAN. */
int hcspmm_plan_check(const hcspmm_plan_header* header_h, int64_t num_nodes, int64_t num_edges,
                      int64_t plan_words_available);

/* 64-bit fingerprint of a CSR graph: an order-independent sum of mixed (index, value) pairs over row_pointers
 * and column_index, so host threads and GPU waves can each add their share.  hcspmm_plan_build stores it in the
 * header; a binding that is handed a plan together with graph tensors it has not seen with that plan computes
 * the device variant once (one small kernel + an 8-byte read-back, then cached by the caller) and refuses a
 * mismatch with HCSPMM_EPLAN: a plan built for a different graph with the same N and E (e.g. the same graph
 * after a LOI reorder) would otherwise silently produce a wrong Z.
 *   fingerprint_out_d : 8-byte device buffer, overwritten (zeroed on `stream` first). */
int hcspmm_graph_fingerprint_host(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                                  int64_t num_edges, uint64_t* fingerprint_out_h);
int hcspmm_graph_fingerprint_device(const int32_t* row_pointers_d, const int32_t* column_index_d, int64_t num_nodes,
                                    int64_t num_edges, uint64_t* fingerprint_out_d, void* stream);

/* Bytes of device workspace hcspmm_forward* needs for this plan and embedding_dim (may be 0). */
size_t hcspmm_workspace_bytes(const hcspmm_plan_header* header_h, int embedding_dim);

/* Wide-task threshold hcspmm_forward will use for this plan and embedding_dim: sparse-path rows
 * (or row segments) with more entries than this are summed by a whole wave -- its 64/L lane groups
 * take alternate chunks and are combined by a fixed shuffle tree (deterministic, but not the
 * sequential CSR order); rows at or below it are summed strictly in CSR order by one lane group,
 * bit-identical to the reference's sparse-row loop (K.cu:1377-1380).  Small graphs get a small
 * threshold (latency-bound: the longest row's dependent-load chain is the kernel time), large
 * graphs a large one (throughput-bound).  Returns INT32_MAX when no task is wide.  header_h == NULL
 * asks about the plan-free kernel (fixed threshold of 64 entries). */
int32_t hcspmm_wide_threshold(const hcspmm_plan_header* header_h, int embedding_dim);

/* ------------------------------------------------------------------------------------------
 * forward: Z = A * X.  Replaces the launchers spmm_forward_plus K.cu:410, spmm_forward_plus_more :457,
 * spmm_forward_plus_fixed32 :500, spmm_forward_plus_fixed64 :553 (bindings B.cpp:194-308; Python names
 * forward / forward_more / forward_fixed32 / forward_fixed64 and the backward* aliases B.cpp:516-523) and
 * the kernels
 * spmm_forward_cuda_kernel_arbi_warps_hybrid_{adaptive,adaptive_more,32,64} K.cu:919-1637.
 * One entry point serves every embedding_dim (the fixed32/fixed64 variants are the same math).
 *
 *   X_d[N*D], Z_d[N*D]            fp32 row-major, Z fully overwritten
 *   row_pointers_d .. hybrid_type_d   the 7 graph tensors of the reference API (device)
 *   plan_d / plan_header_h        plan blob in HBM + host copy of its header, or both NULL:
 *                                 then the plan-free kernel runs (one workgroup per row window,
 *                                 branch on hybrid_type, as the reference does)
 *   workspace_d / workspace_bytes >= hcspmm_workspace_bytes(); may be NULL when that is 0
 *   stream                        hipStream_t (as void*); NULL = the null stream
 * Asynchronous: returns after enqueueing; no host-device synchronisation inside.
 * ---------------------------------------------------------------------------------------- */
int hcspmm_forward(const float* X_d, float* Z_d, const int32_t* row_pointers_d, const int32_t* column_index_d,
                   const int32_t* blockPartition_d, const int32_t* edgeToColumn_d, const int32_t* edgeToRow_d,
                   const int32_t* hybrid_type_d, const int32_t* plan_d, const hcspmm_plan_header* plan_header_h,
                   int64_t num_nodes, int64_t num_edges, int embedding_dim, void* workspace_d,
                   size_t workspace_bytes, void* stream);

/* The same product on strided views: X rows are ldx elements apart, Z rows ldz (both >= embedding_dim),
 * so a column panel of a wider matrix can be read / written in place (the multi-GPU shard gathers X
 * panel by panel and writes each panel's product straight into its slice of Z).
 * x_rows = rows of X (hcspmm_forward: num_nodes -- the reference's graphs are square); with a plan, a launch
 * whose plan gathers beyond it (header num_columns > x_rows) is refused with HCSPMM_EINVAL.  The plan-free
 * kernel reads column_index as it is handed over, like the reference: its caller vouches for the range
 * (hcspmm_preprocess_host checked it when it produced the window tensors). */
int hcspmm_forward_strided(const float* X_d, int64_t x_rows, int64_t ldx, float* Z_d, int64_t ldz, const int32_t* row_pointers_d,
                           const int32_t* column_index_d, const int32_t* blockPartition_d,
                           const int32_t* edgeToColumn_d, const int32_t* edgeToRow_d, const int32_t* hybrid_type_d,
                           const int32_t* plan_d, const hcspmm_plan_header* plan_header_h, int64_t num_nodes,
                           int64_t num_edges, int embedding_dim, void* workspace_d, size_t workspace_bytes,
                           void* stream);

/* Feature element types of hcspmm_forward_typed. */
#define HCSPMM_DTYPE_F32 0
#define HCSPMM_DTYPE_F16 1  /* IEEE binary16 */
#define HCSPMM_DTYPE_BF16 2 /* bfloat16 */

/* The same product for fp32, fp16 or bf16 features (the half-precision variants of the paper's Table VII, p.16;
 * the reference repository ships fp32 only).  X and Z hold `dtype` elements, ldx / ldz count elements.  16-bit
 * rows are gathered as stored (half the bytes of the fp32 path -- the launch is bound by exactly those bytes),
 * widened exactly, summed in fp32 in the same order as the fp32 path, and rounded once (to nearest even) per
 * output element:  Z = round_dtype(fp32 sum).  The workspace is fp32 whatever the dtype
 * (hcspmm_workspace_bytes).  dtype = HCSPMM_DTYPE_F32 is hcspmm_forward_strided. */
int hcspmm_forward_typed(const void* X_d, int64_t x_rows, int64_t ldx, void* Z_d, int64_t ldz, int dtype,
                         const int32_t* row_pointers_d, const int32_t* column_index_d,
                         const int32_t* blockPartition_d, const int32_t* edgeToColumn_d, const int32_t* edgeToRow_d,
                         const int32_t* hybrid_type_d, const int32_t* plan_d, const hcspmm_plan_header* plan_header_h,
                         int64_t num_nodes, int64_t num_edges, int embedding_dim, void* workspace_d,
                         size_t workspace_bytes, void* stream);

/* hcspmm_wide_threshold for a feature type (lanes per row, hence the threshold, depend on the element size). */
int32_t hcspmm_wide_threshold_typed(const hcspmm_plan_header* header_h, int embedding_dim, int dtype);

/* ------------------------------------------------------------------------------------------
 * Fused aggregate + update: out2 = A * X (N x D), out = out2 * weights (N x H), weights row-major
 * D x H with row stride weights_ld_row and column stride weights_ld_col in elements (so a
 * transposed view, GNN_model.py:98,120, needs no copy).  Replaces spmm_forward_plus_fixed32_fused
 * K.cu:596, _fixed64_fused :651, _final_fused :701, _final_fused_64 :759, _GIN_final_fused :810
 * (bindings B.cpp:310-498) and their kernels K.cu:1639-2770.
 * `out_d` may be a caller-owned buffer (forward_final_fused writes the caller's `output`).
 *
 * Three forms, the same results (out2 bit-identical in all of them; out bit-identical in forms 0 and 2).
 * 0, two launches: the hybrid launch (out2 = A * X) followed by one streaming MFMA update launch over all rows.
 * 2, row tiles (fused_rows.hip; the reference keeps a window's aggregate in shared memory and multiplies it in the same
 *   block, K.cu:1807-1837): persistent launches take 16 CONSECUTIVE TASKS of the length-sorted task list (the update does
 *   not care which 16 rows share a tile), or one dense-tile window, sum them exactly as the plain kernel does, write the
 *   rows of out2 from the registers, park the 16 x D tile in a wave-private LDS area and run the update's MFMA chain on
 *   it (weights staged in LDS once per workgroup), so out2 is never read back.  Rows summed by whole waves or in pieces
 *   (wide tasks, split and column-sliced rows) stay in the hybrid launch and are multiplied by a small launch behind the
 *   fix-up pass.  fp32, D in [17, 128] and H <= 32 or 49 ... 64 (widths off the 16-column grid are zero-padded inside the launch), sparse region in one column pass (D < 64, or a
 *   short-row graph, or hcspmm_plan_params.panel_cols < 0); D = 128 is summed in two column chunks of 64 (eight-wave
 *   workgroups, `out` accumulated over the chunks in the same order).  +2 ... +34 % over form 0 wherever out2 is 80 MB or
 *   more at D <= 64, +5 ... +22 % on dense-heavy graphs up to D = 128 with H <= 32 (profiles/r03/ab_fused_rows.log) --
 *   chosen automatically there; opt-in elsewhere (slower on small graphs).
 * 1, in-launch (round 2; plans built with fuse_in_launch = 1, dense-tile windows only): the aggregation runs with
 *   exchanged MFMA operands, which leaves each lane holding one row of the 16 x D tile in the A-operand shape of the
 *   (tile x weights) MFMAs, so the tile goes from the accumulators straight into the update; windows on the sparse-row
 *   path are multiplied by a second launch restricted to them (off_sparse_windows).  128 registers, ~100 MFMAs on the
 *   critical path of latency-bound waves: 0-15 % slower than form 0 here (profiles/r02/ab_fused.log); kept for the record.
 * hcspmm_fused_in_launch() tells which form a (plan, D, H) gets; HCSPMM_FUSED_SINGLE_LAUNCH=0 / 1 / 2 in the environment
 * forces a form for every plan (shapes outside it fall back).
 * ---------------------------------------------------------------------------------------- */
int hcspmm_fused_in_launch(const hcspmm_plan_header* header_h, int embedding_dim, int hidden_dim);
int hcspmm_forward_fused(const float* X_d, float* out_d, float* out2_d, const float* weights_d,
                         int64_t weights_ld_row, int64_t weights_ld_col, int hidden_dim,
                         const int32_t* row_pointers_d, const int32_t* column_index_d,
                         const int32_t* blockPartition_d, const int32_t* edgeToColumn_d,
                         const int32_t* edgeToRow_d, const int32_t* hybrid_type_d, const int32_t* plan_d,
                         const hcspmm_plan_header* plan_header_h, int64_t num_nodes, int64_t num_edges,
                         int embedding_dim, void* workspace_d, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The update GEMM on its own:  out[N x H] = in[N x D] * W  (fp32; in / out row-major and contiguous, W with element
 * strides ldr / ldc, so a transposed view needs no copy).  Replaces torch.mm(X, weights) / torch.mm(d, weights.t()) in the
 * layers (GNN_model.py:67,87,110,134,194 and the backward passes): N is in the millions and D, H a few dozen, so the product
 * is a stream over `in`; the kernel of the fused operators' update launch (W staged in LDS once per workgroup, 16-byte loads,
 * fp32 MFMA) runs it at twice the rate of the library GEMM picked for such shapes on MI355X.  Any D, H >= 1 (shapes outside
 * the streaming kernel's take a plain MFMA kernel).  Deterministic.
 * ---------------------------------------------------------------------------------------- */
int hcspmm_dense_update(const float* in_d, const float* weights_d, int64_t ldr, int64_t ldc, float* out_d, int64_t N, int D,
                        int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradient of the update GEMM, for the autograd glue around the operators:
 *   dW[D x H] = A^T * B,  A = N x D (rows lda elements apart), B = N x H (rows ldb apart), fp32, dW row-major.
 * Replaces torch.mm(X.t(), d_out) of the reference's backward passes (GNN_model.py:79,101,124,160,181,205,230):
 * with K = N in the hundreds of thousands and a tiny output the library GEMM runs one tile over all of K; this
 * splits K over the grid (fp32 MFMA, partials added in a fixed order: deterministic).
 * Supported: ceil(D/16) <= 8, ceil(H/16) <= 4, their product <= 16 (else HCSPMM_EINVAL -- use a library GEMM).
 * ---------------------------------------------------------------------------------------- */
size_t hcspmm_weight_grad_workspace(int64_t N, int D, int H); /* bytes; 0 if the shape is unsupported */
int hcspmm_weight_grad(const float* A_d, int64_t lda, const float* B_d, int64_t ldb, float* dW_d, int64_t N, int D,
                       int H, void* workspace_d, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LOI layout reorder (host).  Replaces reorder_plus_new_direct LOI.cpp:660-805 with the in-CSR
 * construction and output ordering of its main (LOI.cpp:826-841, :873-891): perm_out_h[N] lists
 * old vertex ids in their new order (full 16-row groups first, then short groups, then vertices
 * with no out-edges).  Bit-identical output, without the reference's O(N^2/16) per-group bitmap
 * reallocation (LOI.cpp:695) or its 18 269 000-vertex static limit (LOI.cpp:96).
 *   group_sizes_out_h : optional [N] buffer receiving the group sizes in creation order;
 *   n_groups_out      : optional.
 * ---------------------------------------------------------------------------------------- */
int hcspmm_loi_reorder(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                       int64_t num_edges, int32_t* perm_out_h, int32_t* group_sizes_out_h,
                       int64_t* n_groups_out);

/* The same with the reference's other un-windowed variant selectable: HCSPMM_LOI_NEW restates
 * reorder_plus_new (LOI.cpp:505-658), which finds candidate rows through the column's own out-list
 * (it assumes a symmetric graph); on symmetric input both variants give the same order. */
#define HCSPMM_LOI_NEW_DIRECT 0
#define HCSPMM_LOI_NEW 1
/* The reference's two windowed variants (not called by its main): reorder_plus_direct LOI.cpp:286-484 and
 * reorder_plus LOI.cpp:98-284 -- rows ordered by their smallest column id, a group grown from the next 300 rows of
 * that order.  Bit-identical to the reference compiled with this image's libstdc++ (its row order comes from a
 * non-stable std::sort) on the inputs where the reference is defined: at least 50 rows, no row without entries,
 * columns strictly ascending within a row; anything else is HCSPMM_EINVAL (the reference reads out of bounds). */
#define HCSPMM_LOI_WINDOWED_DIRECT 2
#define HCSPMM_LOI_WINDOWED 3
int hcspmm_loi_reorder_variant(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                               int64_t num_edges, int variant, int32_t* perm_out_h, int32_t* group_sizes_out_h,
                               int64_t* n_groups_out);

/* A RELAXED, parallel LOI reorder for graphs on which the exact one above costs more than the training run it is
 * meant to speed up (Reddit-scale: 6.7 s on one host core against 0.4 s for 200 epochs).  Same group growth, profit,
 * tie rule and output order as reorder_plus_new_direct (LOI.cpp:660-805, :873-891); NOT the reference's permutation:
 *   list_cap : a walk of one column's row list reads at most this many rows behind the list's leading run of placed
 *              rows (0 = 64; < 0 = the whole list, as the reference does); of a member's columns the first 4 * list_cap
 *              new ones are walked (hub rows);
 *   batch    : seeds grown concurrently per round against the placement state of the round's start; a row wanted by
 *              several groups of a round goes to the earliest seed (0 = clamp(num_nodes / 2048, 1, 2048); 1 = one seed
 *              at a time, as the reference does);
 *   threads  : host threads (0 = min(16, HCSPMM_THREADS or the hardware's)).
 * The permutation is a function of (graph, batch, list_cap) only -- not of `threads`, not of timing -- and with
 * batch = 1, list_cap < 0 it equals hcspmm_loi_reorder's bit for bit.  params = NULL: all automatic. */
typedef struct hcspmm_loi_fast_params {
  int32_t batch;
  int32_t list_cap;
  int32_t threads;
  int32_t reserved; /* 0 */
} hcspmm_loi_fast_params;
int hcspmm_loi_reorder_fast(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                            int64_t num_edges, const hcspmm_loi_fast_params* params, int32_t* perm_out_h,
                            int32_t* group_sizes_out_h, int64_t* n_groups_out);

/* Apply a LOI permutation to a CSR graph (the step missing from the reference repository,
 * SURVEY.md section 1 L0): new id of old vertex perm[i] is i; rows AND columns are relabelled,
 * columns re-sorted ascending.  Outputs have the sizes of the inputs. */
int hcspmm_apply_permutation(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes,
                             int64_t num_edges, const int32_t* perm_h, int32_t* row_pointers_out_h,
                             int32_t* column_index_out_h);

#ifdef __cplusplus
}
#endif
#endif /* HCSPMM_H */
