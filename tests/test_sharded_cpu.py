"""Multi-process (gloo, world_size 2, CPU) tests of the row-block shard + all-gather(X) layer.
The local operator injected here is the ORACLE (tests may use it); in production it is the HIP
operator (hcspmm.forward_rect) -- hcspmm.sharded itself contains no compute."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hcspmm import graphs
from hcspmm.sharded import ShardedGraph, ShardedSpMM, partition_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, D, out_dir):
    for p in (ROOT, os.path.join(ROOT, "hc-spmm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rp, col = graphs.powerlaw_graph(1003, 15000, seed=seed)  # N % 16 != 0, unequal blocks
        N = len(rp) - 1
        X = np.random.default_rng(seed).standard_normal((N, D)).astype(np.float32)
        ranges = partition_rows(rp, world)
        g = ShardedGraph(rp, col, ranges, rank)

        def local_spmm(X_full):
            return torch.from_numpy(oracle.spmm_f32(g.row_pointers, g.column_index, X_full.numpy()))

        op = ShardedSpMM(g, local_spmm)
        Z_local = op(torch.from_numpy(X[g.r0:g.r1]))
        want = oracle.spmm_f32(rp, col, X)[g.r0:g.r1]
        ok = np.array_equal(Z_local.numpy(), want)

        # pipelined form: column panels gathered asynchronously, each multiplied into its slice of Z
        def local_spmm_into(X_panel_full, Z_view):
            Z_view.copy_(torch.from_numpy(oracle.spmm_f32(g.row_pointers, g.column_index, X_panel_full.numpy())))

        op2 = ShardedSpMM(g, local_spmm, local_spmm_into=local_spmm_into, n_panels=D // 4)
        ok = ok and np.array_equal(op2(torch.from_numpy(X[g.r0:g.r1])).numpy(), want)
        # preprocess runs per shard on local windows with (remapped) global columns: host side only
        import hcspmm
        outs = hcspmm.preprocess(torch.from_numpy(g.column_index), torch.from_numpy(g.row_pointers), g.n_local,
                                 len(g.column_index), (g.n_local + 15) // 16, num_columns=g.world_size * g.pad_rows)
        want_pre = oracle.preprocess(g.row_pointers, g.column_index)
        ok = ok and all(np.array_equal(a, b.numpy()) for a, b in zip(want_pre, outs[:4]))
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([ok, g.r0, g.r1]))
    finally:
        dist.destroy_process_group()


def test_partition_rows_window_aligned_and_balanced():
    rp, col = graphs.powerlaw_graph(5000, 80000, seed=1)
    for world in (1, 2, 4, 8):
        ranges = partition_rows(rp, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == len(rp) - 1
        for (a, b), (c, d) in zip(ranges[:-1], ranges[1:]):
            assert b == c and b % 16 == 0
        nnz = [rp[b] - rp[a] for a, b in ranges]
        assert max(nnz) <= 1.35 * (sum(nnz) / world) + 4500  # hubs bound the achievable balance


def test_column_remap_round_trip():
    rp, col = graphs.uniform_graph(500, 4000, seed=2)
    ranges = partition_rows(rp, 3)
    g = ShardedGraph(rp, col, ranges, 1)
    # a padded gathered matrix built by hand must present the same rows under the remapped ids
    X = np.arange(500, dtype=np.float32)[:, None] * np.ones((1, 4), np.float32)
    full = np.zeros((3 * g.pad_rows, 4), np.float32)
    for p, (a, b) in enumerate(ranges):
        full[p * g.pad_rows:p * g.pad_rows + (b - a)] = X[a:b]
    e0, e1 = rp[g.r0], rp[g.r1]
    assert np.array_equal(full[g.column_index, 0], X[col[e0:e1], 0])


@pytest.mark.parametrize("world,D", [(2, 8), (2, 32), (3, 16)])
def test_sharded_spmm_gloo(tmp_path, world, D):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 7, D, str(tmp_path)), nprocs=world, join=True)
    covered = []
    for r in range(world):
        ok, r0, r1 = np.load(tmp_path / ("ok_%d.npy" % r))
        assert ok == 1
        covered.append((int(r0), int(r1)))
    assert covered[0][0] == 0 and covered[-1][1] == 1003
    assert all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:]))


def _gpu_worker(rank, world, port, out_dir):
    """Two ranks sharing GPU 0 (gloo collectives, host-staged gather): the REAL HIP operator as the
    local product of the row-block shard, unequal blocks (padding + column remap)."""
    for p in (ROOT, os.path.join(ROOT, "hc-spmm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hcspmm
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        rp, col = graphs.powerlaw_graph(5003, 90000, seed=11, max_degree_frac=0.3)
        N, D = len(rp) - 1, 64
        X = np.random.default_rng(3).standard_normal((N, D)).astype(np.float32)
        g = ShardedGraph(rp, col, partition_rows(rp, world), rank)
        rp_d, col_d = torch.from_numpy(g.row_pointers).to(dev), torch.from_numpy(g.column_index).to(dev)
        outs = hcspmm.preprocess(col_d, rp_d, g.n_local, len(g.column_index), (g.n_local + 15) // 16,
                                 num_columns=g.world_size * g.pad_rows)
        op = ShardedSpMM(g, lambda Xf: hcspmm.forward_rect(Xf, rp_d, col_d, *outs)[0],
                         local_spmm_into=lambda Xf, Zv: hcspmm.forward_into(Xf, Zv, rp_d, col_d, *outs), n_panels=2)
        Z = op(torch.from_numpy(X[g.r0:g.r1]).to(dev)).cpu().numpy()
        e0, e1 = rp[g.r0], rp[g.r1]
        ok, ratio = oracle.check_spmm(Z, (rp[g.r0:g.r1 + 1] - e0).astype(np.int32), col[e0:e1], X)
        np.save(os.path.join(out_dir, "gpu_ok_%d.npy" % rank), np.array([ok, ratio]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_spmm_two_ranks_on_one_gpu(tmp_path):
    assert torch.cuda.is_available()
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, ratio = np.load(tmp_path / ("gpu_ok_%d.npy" % r))
        assert ok == 1, ratio
