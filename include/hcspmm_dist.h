/*
 * hcspmm_dist.h -- C ABI of libhcspmm_dist.so: the row-block shard of the hybrid SpMM over the GPUs of one node
 * WITHOUT PyTorch (the reference is single-GPU, HC-SpMM_main.py:47-49 -- nothing is replaced here, this
 * is the multi-GPU step of hcspmm/sharded.py restated for a host that owns its buffers and its RCCL communicator).
 *
 * libhcspmm_dist.so = libhcspmm.so (include/hcspmm.h) + RCCL (<rccl/rccl.h>); libhcspmm.so itself stays free of any
 * communication library.  One process (or thread) per GPU; rank p owns a window-aligned block of rows: A[rows_p, :]
 * (column ids index the rows of the GATHERED embedding matrix: owner * pad_rows + local row), X[rows_p, :], Z[rows_p, :].
 * One exchange per step: ncclAllGather of the X row blocks over xGMI, in column panels, the product of panel k
 * (hcspmm_forward_typed on strided views) enqueued behind gather k so that it runs under gather k + 1.
 *
 * Layout (what makes a step copy- and allocation-free): panel-major buffers of the caller,
 *   x_pm        [n_panels][pad_rows][w]           this rank's features; rows >= n_local are padding (zero)
 *   gathered_pm [n_panels][world * pad_rows][w]   receive buffers, one per panel
 *   z_pm        [n_panels][n_local][w]            the result, same layout as x_pm: feeds the next aggregation as it is
 * with w = embedding_dim / n_panels (32 fp32 columns = one 128-byte line per row is the kernel's own panel width).
 */
#ifndef HCSPMM_DIST_H
#define HCSPMM_DIST_H

#include "hcspmm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hcspmm_dist_ctx hcspmm_dist_ctx; /* HIP events of one shard; create once, on the device that runs it */

typedef struct hcspmm_dist_step_args {
  void* nccl_comm;      /* ncclComm_t: this rank's communicator over the world_size GPUs (created by the caller) */
  int32_t world_size;
  int32_t n_panels;     /* column panels; embedding_dim % n_panels == 0 */
  int32_t embedding_dim;
  int32_t dtype;        /* HCSPMM_DTYPE_F32 / _F16 / _BF16 */
  int64_t pad_rows;     /* common block height of the gathered matrix (max over ranks of n_local) */
  int64_t n_local;      /* rows of this rank's block */
  int64_t num_edges;    /* stored entries of this rank's block */
  const void* x_pm;
  void* gathered_pm;    /* unused (may be NULL) when world_size == 1 and always_gather == 0 */
  void* z_pm;
  /* this rank's block, preprocessed with num_columns = world_size * pad_rows (hcspmm_preprocess_host, hcspmm_plan_build) */
  const int32_t* row_pointers_d;
  const int32_t* column_index_d;
  const int32_t* blockPartition_d;
  const int32_t* edgeToColumn_d;
  const int32_t* edgeToRow_d;
  const int32_t* hybrid_type_d;
  const int32_t* plan_d;                 /* NULL (with plan_header_h NULL): plan-free kernel */
  const hcspmm_plan_header* plan_header_h;
  void* workspace_d;                     /* >= hcspmm_workspace_bytes(plan_header_h, embedding_dim / n_panels) */
  size_t workspace_bytes;
  void* compute_stream;                  /* hipStream_t the products are enqueued on (the caller's stream) */
  void* comm_stream;                     /* hipStream_t the gathers are enqueued on; must differ from compute_stream for overlap */
  int32_t always_gather;                 /* != 0: take the collective path in a one-rank world as well (tests on a one-GPU box) */
} hcspmm_dist_step_args;

/* Host side of the shard (no GPU, no communicator needed): which rows a rank owns and what its block looks like.
 *
 * hcspmm_dist_partition_rows: contiguous row ranges, boundaries multiples of 16 (a row window is never cut), balanced by
 * stored entries + rows -- ranges_out[2 * p] = first row of rank p, ranges_out[2 * p + 1] = one past its last.
 * hcspmm_dist_extract_block: rank's local CSR -- row_pointers_out[n_local + 1] rebased to 0, column_index_out[e_local]
 * with every global vertex id v replaced by the row of the padded gathered matrix that holds it (owner(v) * pad_rows +
 * v - first row of owner(v)); pad_rows_out = the common block height max_p(n_local_p).  The block is then preprocessed
 * with num_columns = world_size * pad_rows (hcspmm_preprocess_host, hcspmm_plan_build).  The caller sizes the outputs:
 * n_local = ranges[2 * rank + 1] - ranges[2 * rank], e_local = row_pointers[r1] - row_pointers[r0]. */
int hcspmm_dist_partition_rows(const int32_t* row_pointers_h, int64_t num_nodes, int world_size, int64_t* ranges_out);
int hcspmm_dist_extract_block(const int32_t* row_pointers_h, const int32_t* column_index_h, int64_t num_nodes, int world_size,
                              const int64_t* ranges, int rank, int32_t* row_pointers_out, int32_t* column_index_out,
                              int64_t* pad_rows_out);

/* max_panels events for "gather p has landed" + one for "the compute stream has reached this step". */
int hcspmm_dist_create(int max_panels, hcspmm_dist_ctx** ctx_out);
void hcspmm_dist_destroy(hcspmm_dist_ctx* ctx);

/* One SpMM of the shard: Z_local = A[rows_p, :] * all_gather(X_local).  Asynchronous: enqueues n_panels gathers on
 * comm_stream (behind everything already enqueued on compute_stream, so that a gather never overwrites a panel the
 * previous step's product still reads, nor reads features not yet produced) and n_panels products on compute_stream
 * (product p behind gather p).  Returns HCSPMM_OK, an hcspmm.h code, or HCSPMM_EHIP for a HIP / RCCL failure
 * (hcspmm_dist_last_error() holds the ncclResult_t or hipError_t). */
int hcspmm_dist_step(hcspmm_dist_ctx* ctx, const hcspmm_dist_step_args* args);
int hcspmm_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HCSPMM_DIST_H */
