"""GPU tests of the torch extension module `HCSPMM` (hc-spmm_amd/hybrid_kernel), i.e. the reference's
own Python-visible boundary (hybrid_all.cpp:500-525 there), against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from hcspmm import graphs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "hc-spmm_amd", "hybrid_kernel")


@pytest.fixture(scope="module")
def HCSPMM():
    if EXT not in sys.path:
        sys.path.insert(0, EXT)
    import HCSPMM as m  # built in-tree by __graft_entry__.build(); fails loudly if absent
    return m


def test_module_surface_matches_reference(HCSPMM):
    # every name bound in the reference's PYBIND11_MODULE (hybrid_all.cpp:500-525)
    names = ["preprocess", "forward", "forward_more", "forward_fixed32", "forward_fixed32_fused", "forward_final_fused",
             "forward_fixed64", "forward_fixed64_fused", "forward_final_fused_64", "forward_GIN_final_fused", "backward",
             "backward_fixed32", "backward_fixed32_fused", "backward_final_fused", "backward_fixed64",
             "backward_fixed64_fused", "backward_final_fused_64", "backward_GIN_final_fused"]
    for n in names:
        assert callable(getattr(HCSPMM, n)), n
    import HYGNN  # the old module name used by HC-SpMM_main.py:52
    assert HYGNN.preprocess is HCSPMM.preprocess or callable(HYGNN.preprocess)


def test_cpu_input_is_rejected_with_reference_message(HCSPMM):
    rp, col = graphs.powerlaw_graph(100, 400, seed=1)
    outs = HCSPMM.preprocess(torch.from_numpy(col), torch.from_numpy(rp), 100, len(col), 7)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        HCSPMM.forward(torch.zeros(100, 8), torch.from_numpy(rp), torch.from_numpy(col), *outs)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [32, 128, 22])
def test_extension_forward_and_fused(HCSPMM, oracle_mod, D):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    rp, col = graphs.planted_dense_graph(1200, seed=5)
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    HCSPMM.set_plan_params(0, 0, False, -1, 0)  # CSR-order bits asserted for every row below: no column slices
    try:
        outs = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)  # column_index FIRST (HC-SpMM_main.py:52)
    finally:
        HCSPMM.set_plan_params(0, 0)
    assert HCSPMM.get_rule() == 3  # the module's default classifier: the width-agnostic MI355X refit (set_rule(0): the reference's)
    want = oracle_mod.preprocess(rp, col, 3)
    for w, g in zip(want, outs[:4]):
        assert g.is_cuda and g.dtype == torch.int32 and np.array_equal(w, g.cpu().numpy())
    HCSPMM.set_rule(0)
    try:  # the reference's coefficients: hybrid_type as the reference computes it
        ref = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    finally:
        HCSPMM.set_rule(3)
    for w, g in zip(oracle_mod.preprocess(rp, col, oracle_mod.RULE_INTENDED), ref[:4]):
        assert np.array_equal(w, g.cpu().numpy())
    assert int(outs[3].sum()) > 0  # some dense-tile windows
    rng = np.random.default_rng(D)
    X = rng.standard_normal((N, D)).astype(np.float32)
    Xd = torch.from_numpy(X).to(dev)
    ref = oracle_mod.spmm_f32(rp, col, X)
    for fn in (HCSPMM.forward, HCSPMM.forward_fixed32, HCSPMM.backward, HCSPMM.forward_more):
        out = fn(Xd, rp_d, col_d, *outs)
        assert isinstance(out, list) and len(out) == 1
        assert np.array_equal(out[0].cpu().numpy(), ref)
    # reference placeholders for row_nzr / col_nzr -> plan-free kernel, same answer
    ph = torch.zeros(1, dtype=torch.int32, device=dev)
    assert np.array_equal(HCSPMM.forward(Xd, rp_d, col_d, outs[0], outs[1], outs[2], outs[3], ph, ph)[0].cpu().numpy(), ref)
    # a cloned plan tensor (unknown pointer) is recognised by its header
    assert np.array_equal(HCSPMM.forward(Xd, rp_d, col_d, outs[0], outs[1], outs[2], outs[3], outs[4].clone(),
                                         outs[5])[0].cpu().numpy(), ref)
    # half-precision features (paper Table VII): same module functions, Z in the input's dtype
    for dt in (torch.float16, torch.bfloat16):
        X16 = Xd.to(dt)
        out16 = HCSPMM.forward(X16, rp_d, col_d, *outs)[0]
        assert out16.dtype == dt
        want16 = torch.from_numpy(oracle_mod.spmm_f32(rp, col, X16.float().cpu().numpy())).to(dt)
        assert torch.equal(out16.cpu().view(torch.int16), want16.view(torch.int16))  # no wide / split rows in this graph
    H = 16
    W = rng.standard_normal((D, H)).astype(np.float32)
    Wd = torch.from_numpy(W).to(dev)
    want_out, want_out2 = oracle_mod.spmm_fused_f32(rp, col, X, W)
    scale = oracle_mod.spmm_f64(rp, col, X, absolute=True) @ np.abs(W).astype(np.float64)  # sum|x_j| . |W|
    out, out2 = HCSPMM.forward_fixed32_fused(Xd, rp_d, col_d, *outs, Wd)
    assert oracle_mod.check_spmm(out2.cpu().numpy(), rp, col, X)[0]
    assert np.all(np.abs(out.cpu().numpy() - want_out) <= 1e-5 * scale + 1e-30)
    buf = torch.zeros(N, H, device=dev)
    out, _ = HCSPMM.forward_final_fused(Xd, rp_d, col_d, *outs, Wd.t().contiguous().t(), buf)
    assert out.data_ptr() == buf.data_ptr()
    assert np.all(np.abs(buf.cpu().numpy() - want_out) <= 1e-5 * scale + 1e-30)


@pytest.mark.gpu
def test_extension_as_shipped_rule_and_side_stream(HCSPMM, oracle_mod):
    dev = torch.device("cuda:0")
    rp, col = graphs.planted_dense_graph(640, seed=6)
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    HCSPMM.set_rule(2)
    HCSPMM.set_plan_params(0, 0, False, -1, 0)  # CSR-order bits asserted for every row below: no column slices (explicit beats HCSPMM_SLICE_THRESHOLD)
    try:
        outs = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    finally:
        HCSPMM.set_rule(3)
        HCSPMM.set_plan_params(0, 0)
    assert int(outs[3].sum()) == 0  # hybrid_all_kernel.cu:262 as shipped: every window sparse
    X = np.random.default_rng(0).standard_normal((N, 64)).astype(np.float32)
    Xd = torch.from_numpy(X).to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):  # launches follow torch's current stream
        Z = HCSPMM.forward(Xd, rp_d, col_d, *outs)[0]
    s.synchronize()
    assert np.array_equal(Z.cpu().numpy(), oracle_mod.spmm_f32(rp, col, X))


def test_extension_host_utilities(HCSPMM):
    """LOI reorder / permutation / plan_info through the compiled module (host side, no GPU needed)."""
    import glob
    gold = os.path.join(ROOT, "tests", "golden")
    for path in sorted(glob.glob(os.path.join(gold, "loi_win_*.npz"))):  # the windowed variants (2 / 3)
        g = np.load(path)
        rp, col = torch.from_numpy(g["row_pointers"]), torch.from_numpy(g["column_index"])
        for variant, key in ((2, "plus_direct"), (3, "plus")):
            perm, sizes = HCSPMM.loi_reorder(rp, col, variant)
            assert np.array_equal(perm.numpy(), g["order_" + key]) and np.array_equal(sizes.numpy(), g["group_sizes_" + key])
    for path in sorted(glob.glob(os.path.join(gold, "loi_*.npz"))):
        if "loi_win_" in path:
            continue
        g = np.load(path)
        rp, col = torch.from_numpy(g["row_pointers"]), torch.from_numpy(g["column_index"])
        perm, sizes = HCSPMM.loi_reorder(rp, col)
        assert np.array_equal(perm.numpy(), g["order"]) and np.array_equal(sizes.numpy(), g["group_sizes"])
        perm_n, _ = HCSPMM.loi_reorder(rp, col, 1)
        assert np.array_equal(perm_n.numpy(), g["order_new"])
        perm_f, sizes_f = HCSPMM.loi_reorder_fast(rp, col, batch=1, list_cap=-1)  # both relaxations off: the reference's order
        assert np.array_equal(perm_f.numpy(), g["order"]) and np.array_equal(sizes_f.numpy(), g["group_sizes"])
        rp2, col2 = HCSPMM.apply_permutation(rp, col, perm)
        assert rp2.numel() == rp.numel() and int(rp2[-1]) == col.numel() and col2.numel() == col.numel()
    rp, col, _ = graphs.community_graph(40000, 84000, seed=4)
    import hcspmm
    a = HCSPMM.loi_reorder_fast(torch.from_numpy(rp), torch.from_numpy(col))
    b = hcspmm.loi_reorder(torch.from_numpy(rp), torch.from_numpy(col), variant="fast", threads=3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    rp, col = graphs.planted_dense_graph(400, seed=3)
    outs = HCSPMM.preprocess(torch.from_numpy(col), torch.from_numpy(rp), 400, len(col), 25)
    info = HCSPMM.plan_info(outs[4])
    assert info["n_dense"] == int(outs[3].sum()) and info["nnz_sparse"] + info["nnz_dense"] == len(col)
    assert HCSPMM.plan_info(torch.zeros(1, dtype=torch.int32)) == {}


@pytest.mark.gpu
def test_extension_weight_grad_and_layer_backward_use_it(HCSPMM, oracle_mod):
    """HCSPMM.weight_grad (dW = A^T B, split-K MFMA kernel) and the layers' backward pass on a graph large enough
    (>= 4096 nodes) for GNN_model._weight_grad to take it: gradients against dense autograd."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    A, B = torch.randn(6000, 96, device=dev), torch.randn(6000, 22, device=dev)
    got = HCSPMM.weight_grad(A, B)
    ref = A.double().t() @ B.double()
    scale = A.double().abs().t() @ B.double().abs()
    assert got.shape == (96, 22) and bool(((got.double() - ref).abs() <= 1e-5 * scale).all())
    assert HCSPMM.weight_grad(torch.zeros(10, 200, device=dev), torch.zeros(10, 8, device=dev)) is None
    import GNN_model
    rp, col = graphs.powerlaw_graph(5000, 40000, seed=12)
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    outs = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    X = torch.randn(N, 32, device=dev, requires_grad=True)
    W = torch.randn(32, 16, device=dev, requires_grad=True)
    Y = GNN_model.HCSPMMFunction.apply(X, W, rp_d, col_d, *outs)
    G = torch.randn_like(Y)
    Y.backward(G)
    Ad = torch.zeros(N, N, dtype=torch.float64, device=dev)
    Ad[torch.from_numpy(np.repeat(np.arange(N), np.diff(rp))).to(dev), col_d.long()] = 1.0
    Xd, Wd = X.detach().double().requires_grad_(), W.detach().double().requires_grad_()
    (Ad @ (Xd @ Wd)).backward(G.double())
    assert torch.allclose(W.grad.double(), Wd.grad, rtol=1e-4, atol=1e-4 * float(Wd.grad.abs().max()))
    assert torch.allclose(X.grad.double(), Xd.grad, rtol=1e-4, atol=1e-4 * float(Xd.grad.abs().max()))


@pytest.mark.gpu
def test_extension_plan_params_select_the_in_launch_fused_form(HCSPMM, oracle_mod):
    """HCSPMM.set_plan_params(split, segment, fuse_in_launch): plans made by the following preprocess calls ask the fused
    operators to update dense-tile windows inside the hybrid launch; the default (and a reset) is the two-launch form.
    Same results either way."""
    dev = torch.device("cuda:0")
    rp, col = graphs.planted_dense_graph(1000, seed=8)
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, 64)).astype(np.float32)
    W = rng.standard_normal((64, 32)).astype(np.float32)
    Xd, Wd = torch.from_numpy(X).to(dev), torch.from_numpy(W).to(dev)
    want_out, _ = oracle_mod.spmm_fused_f32(rp, col, X, W)
    scale = oracle_mod.spmm_f64(rp, col, X, absolute=True) @ np.abs(W).astype(np.float64)
    outs = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    assert not HCSPMM.fused_in_launch(outs[4], 64, 32) and HCSPMM.plan_info(outs[4])["flags"] == 0
    ref_out, ref_out2 = HCSPMM.forward_fixed32_fused(Xd, rp_d, col_d, *outs, Wd)
    HCSPMM.set_plan_params(0, 0, True)
    try:
        outs2 = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    finally:
        HCSPMM.set_plan_params(0, 0)
    assert HCSPMM.plan_info(outs2[4])["flags"] == 1 and HCSPMM.plan_info(outs2[4])["n_dense"] > 0
    assert HCSPMM.fused_in_launch(outs2[4], 64, 32) and not HCSPMM.fused_in_launch(outs2[4], 64, 64)
    out, out2 = HCSPMM.forward_fixed32_fused(Xd, rp_d, col_d, *outs2, Wd)
    assert torch.equal(out2, ref_out2)
    for o in (out, ref_out):
        assert np.all(np.abs(o.cpu().numpy().astype(np.float64) - want_out) <= 1e-5 * scale + 1e-30)
    outs3 = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    assert HCSPMM.plan_info(outs3[4])["flags"] == 0
    # fuse_in_launch = 2: the row-tile form (sparse rows as well); `out` then has the two-launch form's bits
    HCSPMM.set_plan_params(0, 0, 2, 0, 0, -1)
    try:
        outs4 = HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16)
    finally:
        HCSPMM.set_plan_params(0, 0)
    assert HCSPMM.plan_info(outs4[4])["flags"] == 2 and HCSPMM.fused_in_launch(outs4[4], 64, 32) == 2
    out, out2 = HCSPMM.forward_fixed32_fused(Xd, rp_d, col_d, *outs4, Wd)
    assert torch.equal(out2, ref_out2) and torch.equal(out, ref_out)
