// hybrid_all.cpp -- PyTorch-ROCm extension module `HCSPMM`: the operator boundary of the
// reference (hybrid_kernel/hybrid_all.cpp:500-525 there) re-hosted over the C ABI of
// libhcspmm.so (include/hcspmm.h).  Same Python-visible names, arity, argument order, return
// lists and CHECK_INPUT messages (reference :185-187), so GNN_model.py / HC-SpMM_main.py written
// against the reference import and call it unchanged.  Host-only translation unit: no kernels,
// no hipify, no CUDA headers -- tensors are used for device memory and the current stream only.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hcspmm.h"

namespace {

#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
  CHECK_CUDA(x);       \
  CHECK_CONTIGUOUS(x)

int g_rule = HCSPMM_RULE_INTENDED;
hcspmm_plan_params g_params = {0, 0};

void check_rc(int rc, const char* what) {
  TORCH_CHECK(rc == HCSPMM_OK, "HCSPMM.", what, ": ", hcspmm_strerror(rc), " [code ", rc, ", hipError_t ",
              hcspmm_last_hip_error(), "]");
}

// plan registry: device pointer of a plan tensor -> (tensor kept alive, host copy of the header).
struct PlanEntry {
  torch::Tensor keep;
  hcspmm_plan_header header;
};
std::mutex g_mu;
std::unordered_map<const void*, PlanEntry> g_plans;

void remember(const torch::Tensor& plan, const hcspmm_plan_header& h) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_plans.size() >= 256) g_plans.erase(g_plans.begin());
  g_plans[plan.data_ptr()] = PlanEntry{plan, h};
}

// Returns true and fills *h when `row_nzr` carries a plan; false for the reference's [0] placeholder.
bool lookup(const torch::Tensor& row_nzr, int64_t N, int64_t E, hcspmm_plan_header* h) {
  if (!row_nzr.defined() || !row_nzr.is_cuda() || row_nzr.scalar_type() != torch::kInt ||
      row_nzr.numel() < HCSPMM_PLAN_HEADER_WORDS || !row_nzr.is_contiguous())
    return false;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(row_nzr.data_ptr());
    if (it != g_plans.end()) {
      *h = it->second.header;
      return true;
    }
  }
  // first sight of this tensor (e.g. a clone): one small device read
  auto host = row_nzr.slice(0, 0, HCSPMM_PLAN_HEADER_WORDS).cpu().contiguous();
  std::memcpy(h, host.data_ptr<int>(), sizeof(*h));
  if (hcspmm_plan_check(h, N, E) != HCSPMM_OK) return false;
  remember(row_nzr, *h);
  return true;
}

struct Call {
  int64_t N, E;
  int D;
  bool has_plan;
  hcspmm_plan_header header;
  torch::Tensor workspace;
  void* stream;
};

// HCSPMM_DTYPE_* of a feature tensor, -1 if unsupported
int feature_dtype(const torch::Tensor& t) {
  switch (t.scalar_type()) {
    case torch::kFloat: return HCSPMM_DTYPE_F32;
    case torch::kHalf: return HCSPMM_DTYPE_F16;
    case torch::kBFloat16: return HCSPMM_DTYPE_BF16;
    default: return -1;
  }
}

Call prepare(const torch::Tensor& input, const torch::Tensor& nodePointer, const torch::Tensor& edgeList,
             const torch::Tensor& blockPartition, const torch::Tensor& edgeToColumn, const torch::Tensor& edgeToRow,
             const torch::Tensor& row_nzr, bool allow_16bit = false) {
  CHECK_INPUT(input);
  CHECK_INPUT(nodePointer);
  CHECK_INPUT(edgeList);
  CHECK_INPUT(blockPartition);
  CHECK_INPUT(edgeToColumn);
  CHECK_INPUT(edgeToRow);
  TORCH_CHECK(input.dim() == 2 && (allow_16bit ? feature_dtype(input) >= 0 : input.scalar_type() == torch::kFloat),
              allow_16bit ? "input must be a 2-D float32 / float16 / bfloat16 tensor" : "input must be a 2-D float32 tensor");
  TORCH_CHECK(nodePointer.scalar_type() == torch::kInt && edgeList.scalar_type() == torch::kInt,
              "nodePointer / edgeList must be int32");
  Call c;
  c.N = nodePointer.size(0) - 1;  // reference :212-214
  c.E = edgeList.size(0);
  c.D = (int)input.size(1);
  TORCH_CHECK(input.size(0) == c.N, "input has ", input.size(0), " rows but the graph has ", c.N, " nodes");
  c.has_plan = lookup(row_nzr, c.N, c.E, &c.header);
  if (c.has_plan) {
    const size_t need = hcspmm_workspace_bytes(&c.header, c.D);
    if (need) c.workspace = torch::empty({(int64_t)(need / 4)}, input.options().dtype(torch::kFloat));
  }
  c.stream = (void*)c10::hip::getCurrentHIPStream(input.device().index()).stream();
  return c;
}

const int* iptr(const torch::Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<int>() : nullptr; }
int* mptr(torch::Tensor& t) { return t.numel() > 0 ? t.data_ptr<int>() : nullptr; }

torch::Tensor run_spmm(const torch::Tensor& input, const torch::Tensor& nodePointer, const torch::Tensor& edgeList,
                       const torch::Tensor& blockPartition, const torch::Tensor& edgeToColumn,
                       const torch::Tensor& edgeToRow, const torch::Tensor& hybrid_type,
                       const torch::Tensor& row_nzr) {
  // fp16 / bf16 features too (the paper's half-precision variants, Table VII): Z has the input's dtype
  Call c = prepare(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, row_nzr, true);
  auto output = torch::empty({c.N, (int64_t)c.D}, input.options());  // reference K.cu:431-433
  const c10::DeviceGuard guard(input.device());
  const int rc = hcspmm_forward_typed(
      input.data_ptr(), c.D, output.data_ptr(), c.D, feature_dtype(input), iptr(nodePointer), iptr(edgeList),
      iptr(blockPartition), iptr(edgeToColumn), iptr(edgeToRow), iptr(hybrid_type), c.has_plan ? iptr(row_nzr) : nullptr,
      c.has_plan ? &c.header : nullptr, c.N, c.E, c.D, c.workspace.defined() ? c.workspace.data_ptr() : nullptr,
      c.workspace.defined() ? (size_t)c.workspace.numel() * 4 : 0, c.stream);
  check_rc(rc, "forward");
  return output;
}

std::vector<torch::Tensor> run_fused(const torch::Tensor& input, const torch::Tensor& nodePointer,
                                     const torch::Tensor& edgeList, const torch::Tensor& blockPartition,
                                     const torch::Tensor& edgeToColumn, const torch::Tensor& edgeToRow,
                                     const torch::Tensor& hybrid_type, const torch::Tensor& row_nzr,
                                     const torch::Tensor& weights, torch::Tensor output) {
  Call c = prepare(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, row_nzr);
  // The reference takes weights.data<float>() without CHECK_INPUT and therefore multiplies a
  // transposed view by its raw storage; here the view's strides are honoured.
  TORCH_CHECK(weights.is_cuda() && weights.scalar_type() == torch::kFloat && weights.dim() == 2 &&
                  weights.size(0) == c.D,
              "weights must be a CUDA float32 tensor of shape [embedding_dim, hidden_dim]");
  const int H = (int)weights.size(1);
  if (!output.defined()) {
    output = torch::empty({c.N, (int64_t)H}, input.options());
  } else {
    CHECK_INPUT(output);
    TORCH_CHECK(output.scalar_type() == torch::kFloat && output.numel() == c.N * H,
                "output must be float32 with num_nodes*hidden_dim elements");
  }
  auto output2 = torch::empty({c.N, (int64_t)c.D}, input.options());
  const c10::DeviceGuard guard(input.device());
  const int rc = hcspmm_forward_fused(
      input.data_ptr<float>(), output.data_ptr<float>(), output2.data_ptr<float>(), weights.data_ptr<float>(),
      weights.stride(0), weights.stride(1), H, iptr(nodePointer), iptr(edgeList), iptr(blockPartition),
      iptr(edgeToColumn), iptr(edgeToRow), iptr(hybrid_type), c.has_plan ? iptr(row_nzr) : nullptr,
      c.has_plan ? &c.header : nullptr, c.N, c.E, c.D, c.workspace.defined() ? c.workspace.data_ptr() : nullptr,
      c.workspace.defined() ? (size_t)c.workspace.numel() * 4 : 0, c.stream);
  check_rc(rc, "forward_fused");
  return {output, output2};
}

}  // namespace

// preprocess(column_index, row_pointers, num_nodes, num_edges, num_row_windows)  -- reference
// hybrid_all.cpp:13-17 / hybrid_all_kernel.cu:339-408; note column_index comes FIRST
// (HC-SpMM_main.py:52).  Host-side (north_star); outputs live on the device of the inputs.
std::vector<torch::Tensor> preprocess(torch::Tensor edgeList_tensor, torch::Tensor nodePointer_tensor, int num_nodes,
                                      int edge_num, int block_num) {
  (void)edge_num;  // the reference's count is the raw line count (dataset.py:59); the tensor size is used
  auto dev = edgeList_tensor.device();
  // large device arrays come back through a pinned buffer (a pageable D2H copy runs at ~1.4 GB/s)
  auto to_host = [](const torch::Tensor& t) {
    if (t.is_cuda() && t.scalar_type() == torch::kInt && t.is_contiguous() && t.numel() > (1 << 16)) {
      auto host = torch::empty(t.sizes(), torch::TensorOptions().dtype(torch::kInt).pinned_memory(true));
      host.copy_(t, /*non_blocking=*/true);
      c10::hip::getCurrentHIPStream(t.device().index()).synchronize();
      return host;
    }
    return t.to(torch::kCPU, torch::kInt).contiguous();
  };
  auto col = to_host(edgeList_tensor);
  auto rp = to_host(nodePointer_tensor);
  const int64_t N = rp.numel() - 1, E = col.numel(), W = (N + HCSPMM_BLK_H - 1) / HCSPMM_BLK_H;
  TORCH_CHECK(num_nodes == N, "preprocess: num_nodes (", num_nodes, ") != row_pointers.size(0)-1 (", N, ")");
  TORCH_CHECK(block_num == W, "preprocess: num_row_windows (", block_num, ") != ceil(N/16) (", W, ")");
  auto opts = torch::TensorOptions().dtype(torch::kInt);
  auto bp = torch::empty({W}, opts), ht = torch::empty({W}, opts), e2c = torch::empty({E}, opts);
  // edgeToRow is the plain CSR row expansion (reference fill_edgeToRow, K.cu:314-326): made on the
  // device when the graph lives there, so 4*E bytes skip the host round trip
  torch::Tensor e2r;
  if (dev.is_cuda()) {
    auto rp64 = nodePointer_tensor.to(dev, torch::kLong);
    e2r = torch::repeat_interleave(torch::arange(N, torch::TensorOptions().dtype(torch::kInt).device(dev)),
                                   rp64.slice(0, 1, N + 1) - rp64.slice(0, 0, N), /*dim=*/c10::nullopt, /*output_size=*/E);
  } else {
    e2r = torch::empty({E}, opts);
  }
  check_rc(hcspmm_preprocess_host(rp.data_ptr<int>(), iptr(col), N, E, g_rule, 0, mptr(bp), mptr(e2c),
                                  dev.is_cuda() ? nullptr : mptr(e2r), mptr(ht)),
           "preprocess");
  int64_t words = 0;
  check_rc(hcspmm_plan_words(rp.data_ptr<int>(), N, E, iptr(bp), iptr(ht), &g_params, &words), "preprocess(plan size)");
  auto plan = torch::empty({std::max<int64_t>(words, HCSPMM_PLAN_HEADER_WORDS)}, opts);
  check_rc(hcspmm_plan_build(rp.data_ptr<int>(), iptr(col), N, E, iptr(bp), iptr(e2c), iptr(ht), &g_params,
                             plan.data_ptr<int>(), plan.numel()),
           "preprocess(plan build)");
  hcspmm_plan_header h;
  std::memcpy(&h, plan.data_ptr<int>(), sizeof(h));
  auto plan_d = plan.to(dev);
  if (plan_d.is_cuda()) remember(plan_d, h);
  auto col_nzr = torch::zeros({1}, opts).to(dev);  // stays the reference's placeholder (K.cu:405)
  return {bp.to(dev), e2c.to(dev), e2r.to(dev), ht.to(dev), plan_d, col_nzr};
}

#define HCSPMM_GRAPH_PARAMS                                                                                   \
  torch::Tensor input, torch::Tensor nodePointer, torch::Tensor edgeList, torch::Tensor blockPartition,       \
      torch::Tensor edgeToColumn, torch::Tensor edgeToRow, torch::Tensor hybrid_type, torch::Tensor row_nzr,  \
      torch::Tensor col_nzr

// reference hybrid_all.cpp:194-308: forward / forward_more / forward_fixed32 / forward_fixed64 all
// compute A*X; one implementation serves every embedding_dim.
std::vector<torch::Tensor> spmm_forward(HCSPMM_GRAPH_PARAMS) {
  return {run_spmm(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr)};
}

// reference hybrid_all.cpp:310-370, :469-498: -> {(A*X)*weights, A*X}
std::vector<torch::Tensor> spmm_forward_fused(HCSPMM_GRAPH_PARAMS, torch::Tensor weights) {
  return run_fused(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                   weights, torch::Tensor());
}

// reference hybrid_all.cpp:405-467: writes the caller's `output` and returns {output, A*X}
std::vector<torch::Tensor> spmm_forward_final_fused(HCSPMM_GRAPH_PARAMS, torch::Tensor weights,
                                                    torch::Tensor output) {
  return run_fused(input, nodePointer, edgeList, blockPartition, edgeToColumn, edgeToRow, hybrid_type, row_nzr,
                   weights, output);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("preprocess", &preprocess, "Preprocess step: window condensing + classifier + MI355X launch plan (host)");
  // forward computation (names of reference hybrid_all.cpp:504-512)
  m.def("forward", &spmm_forward, "HCSPMM SPMM forward (gfx950)");
  m.def("forward_more", &spmm_forward, "HCSPMM SPMM forward more (gfx950)");
  m.def("forward_fixed32", &spmm_forward, "HCSPMM SPMM forward fixed32 (gfx950)");
  m.def("forward_fixed32_fused", &spmm_forward_fused, "HCSPMM SPMM forward fixed32 fused (gfx950)");
  m.def("forward_final_fused", &spmm_forward_final_fused, "HCSPMM SPMM forward final fused (gfx950)");
  m.def("forward_fixed64", &spmm_forward, "HCSPMM SPMM forward fixed64 (gfx950)");
  m.def("forward_fixed64_fused", &spmm_forward_fused, "HCSPMM SPMM forward fixed64 fused (gfx950)");
  m.def("forward_final_fused_64", &spmm_forward_final_fused, "HCSPMM SPMM forward final fused 64 (gfx950)");
  m.def("forward_GIN_final_fused", &spmm_forward_fused, "HCSPMM SPMM forward for GIN final fused (gfx950)");
  // backward: the reference binds every backward name to the forward function (:516-523)
  m.def("backward", &spmm_forward, "HCSPMM SPMM backward (gfx950)");
  m.def("backward_fixed32", &spmm_forward, "HCSPMM SPMM backward fixed32 (gfx950)");
  m.def("backward_fixed32_fused", &spmm_forward_fused, "HCSPMM SPMM backward fixed32 fused (gfx950)");
  m.def("backward_final_fused", &spmm_forward_final_fused, "HCSPMM SPMM backward final fused (gfx950)");
  m.def("backward_fixed64", &spmm_forward, "HCSPMM SPMM backward fixed 64 (gfx950)");
  m.def("backward_fixed64_fused", &spmm_forward_fused, "HCSPMM SPMM backward fixed 64 fused (gfx950)");
  m.def("backward_final_fused_64", &spmm_forward_final_fused, "HCSPMM SPMM backward final fused 64 (gfx950)");
  m.def("backward_GIN_final_fused", &spmm_forward_fused, "HCSPMM SPMM backward for GIN final fused (gfx950)");
  // additions (not in the reference): classifier rule / plan tunables, LOI reorder on the host
  m.def("set_rule", [](int rule) {
    TORCH_CHECK(rule >= HCSPMM_RULE_INTENDED && rule <= HCSPMM_RULE_MI355X_WIDE, "unknown rule");
    g_rule = rule;
  }, "0 = intended classifier (default), 1 = with the size>32 guard, 2 = as shipped (hybrid_all_kernel.cu:262), 3 = MI355X refit");
  m.def("set_plan_params", [](int split_threshold, int segment_len) {
    g_params.split_threshold = split_threshold;
    g_params.segment_len = segment_len;
  }, "rows longer than split_threshold are cut into segments of segment_len entries (0 = defaults)");
  m.def("abi_version", []() { return hcspmm_abi_version(); });
  // LOI layout reorder on the host (the reference ships it as a separate file-to-file program, LOI.cpp)
  m.def("loi_reorder", [](torch::Tensor row_pointers, torch::Tensor column_index, int variant) {
    auto rp = row_pointers.to(torch::kCPU, torch::kInt).contiguous();
    auto col = column_index.to(torch::kCPU, torch::kInt).contiguous();
    const int64_t N = rp.numel() - 1, E = col.numel();
    auto perm = torch::empty({N}, torch::kInt), sizes = torch::empty({std::max<int64_t>(N, 1)}, torch::kInt);
    int64_t ng = 0;
    check_rc(hcspmm_loi_reorder_variant(rp.data_ptr<int>(), iptr(col), N, E, variant, mptr(perm), sizes.data_ptr<int>(), &ng),
             "loi_reorder");
    return std::vector<torch::Tensor>{perm, sizes.slice(0, 0, ng).clone()};
  }, "-> [perm (old vertex id at each new position), group sizes]; variant 0 = reorder_plus_new_direct, 1 = reorder_plus_new",
        pybind11::arg("row_pointers"), pybind11::arg("column_index"), pybind11::arg("variant") = 0);
  m.def("apply_permutation", [](torch::Tensor row_pointers, torch::Tensor column_index, torch::Tensor perm) {
    auto rp = row_pointers.to(torch::kCPU, torch::kInt).contiguous();
    auto col = column_index.to(torch::kCPU, torch::kInt).contiguous();
    auto p = perm.to(torch::kCPU, torch::kInt).contiguous();
    const int64_t N = rp.numel() - 1, E = col.numel();
    TORCH_CHECK(p.numel() == N, "perm must have num_nodes entries");
    auto rp2 = torch::empty({N + 1}, torch::kInt), col2 = torch::empty({E}, torch::kInt);
    check_rc(hcspmm_apply_permutation(rp.data_ptr<int>(), iptr(col), N, E, iptr(p), rp2.data_ptr<int>(), mptr(col2)),
             "apply_permutation");
    return std::vector<torch::Tensor>{rp2, col2};
  }, "relabel a CSR graph with a LOI permutation -> [row_pointers, column_index]");
  m.def("weight_grad", [](torch::Tensor A, torch::Tensor B) -> torch::Tensor {
    // dW = A^T B with K = number of nodes (hcspmm.h hcspmm_weight_grad); an undefined tensor (None) when the
    // shape is outside the kernel's range, so that the caller can use a library GEMM
    if (!(A.is_cuda() && B.is_cuda() && A.scalar_type() == torch::kFloat && B.scalar_type() == torch::kFloat &&
          A.dim() == 2 && B.dim() == 2 && A.size(0) == B.size(0) && A.stride(1) == 1 && B.stride(1) == 1))
      return torch::Tensor();
    const int64_t N = A.size(0);
    const int D = (int)A.size(1), H = (int)B.size(1);
    const size_t need = hcspmm_weight_grad_workspace(N, D, H);
    if (need == 0) return torch::Tensor();
    auto ws = torch::empty({(int64_t)(need / 4)}, A.options());
    auto out = torch::empty({(int64_t)D, (int64_t)H}, A.options());
    const c10::DeviceGuard guard(A.device());
    check_rc(hcspmm_weight_grad(A.data_ptr<float>(), A.stride(0), B.data_ptr<float>(), B.stride(0), out.data_ptr<float>(), N,
                                D, H, ws.data_ptr(), need, (void*)c10::hip::getCurrentHIPStream(A.device().index()).stream()),
             "weight_grad");
    return out;
  }, "dW = A^T B for the layers' backward passes (split-K MFMA kernel); None if the shape is unsupported");
  m.def("plan_info", [](torch::Tensor row_nzr) {
    pybind11::dict d;
    hcspmm_plan_header h;
    if (!row_nzr.defined() || row_nzr.numel() < HCSPMM_PLAN_HEADER_WORDS || row_nzr.scalar_type() != torch::kInt) return d;
    auto host = row_nzr.slice(0, 0, HCSPMM_PLAN_HEADER_WORDS).cpu().contiguous();
    std::memcpy(&h, host.data_ptr<int>(), sizeof(h));
    if (h.magic != HCSPMM_PLAN_MAGIC) return d;
    d["n_tasks"] = h.n_tasks; d["n_dense"] = h.n_dense; d["n_split_rows"] = h.n_split_rows;
    d["n_partials"] = h.n_partials; d["nnz_sparse"] = h.nnz_sparse; d["nnz_dense"] = h.nnz_dense;
    d["uniq_dense"] = h.uniq_dense; d["split_threshold"] = h.split_threshold; d["segment_len"] = h.segment_len;
    return d;
  }, "fields of the launch plan carried in row_nzr ({} for the reference's [0] placeholder)");
}
