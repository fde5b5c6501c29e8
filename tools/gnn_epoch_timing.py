"""Times the callers either side of the hot path on MI355X: per-call host overhead of the HCSPMM
extension, and GCN / GIN training epochs (reference defaults: 6 layers, dim 96, hidden 32, 22 classes,
HC-SpMM_main.py:19-25) on the Reddit-scale synthetic graph, built in memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hc-spmm_amd")
sys.path[:0] = [ROOT, PKG, os.path.join(PKG, "hybrid_kernel")]
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
import HCSPMM
from GNN_model import GCNConv, GINConv, SAG
from hcspmm import graphs

dev = torch.device("cuda:0")


class Data:
    pass


def build(n, e, dim, classes, seed, workload=None):
    if workload:  # one of bench.py's graphs (the paper's Table-II-sized low-degree shapes)
        import bench
        rp, col = bench.make_local_block(workload, n, e, 1, 0)
    else:
        rp, col = graphs.powerlaw_graph(n, e, seed=seed)
    d = Data()
    d.num_nodes, d.num_features, d.num_classes = n, dim, classes
    d.row_pointers, d.column_index = torch.from_numpy(rp).to(dev), torch.from_numpy(col).to(dev)
    d.x = torch.randn(n, dim, device=dev)
    d.y = torch.ones(n, dtype=torch.long, device=dev)
    return d


class Net(nn.Module):
    def __init__(self, conv, d, graph, output, hidden, layers):
        super().__init__()
        self.d, self.graph, self.output = d, graph, output
        self.conv1 = conv(d.num_features, hidden, 1)
        self.hidden_layers = nn.ModuleList(conv(hidden, hidden, 0) for _ in range(layers - 2))
        self.conv2 = conv(hidden, d.num_classes, 2)

    def forward(self):
        x = F.relu(self.conv1(self.d.x, *self.graph, self.output))
        x = F.dropout(x, training=self.training)
        for c in self.hidden_layers:
            x = F.relu(c(x, *self.graph, self.output))
        return F.log_softmax(self.conv2(x, *self.graph, self.output), dim=1)


# usage: gnn_epoch_timing.py [rd_like|yh_like|tt_like]   (default: Cora- and Reddit-scale); HCSPMM_FUSED_SINGLE_LAUNCH=0 in the
# environment gives the two-launch form of the fused operators for an A/B of the million-row graphs
CASES = (("cora-scale", 10000, 50000, None), ("reddit-scale", 233000, 11600000, None))
if len(sys.argv) > 1:
    import bench
    CASES = tuple((w + " (bench.py workload)", bench.WORKLOADS[w][0], bench.WORKLOADS[w][1], w) for w in sys.argv[1:])
for name, n, e, wl in CASES:
    d = build(n, e, 96, 22, 1 if n == 10000 else 3, wl)
    t0 = time.perf_counter()
    outs = HCSPMM.preprocess(d.column_index, d.row_pointers, n, d.column_index.numel(), (n + 15) // 16)
    torch.cuda.synchronize()
    print("%s: preprocess %.1f ms" % (name, (time.perf_counter() - t0) * 1e3))
    graph = (d.row_pointers, d.column_index, *outs)
    X32 = torch.randn(n, 32, device=dev)
    for _ in range(20):
        HCSPMM.forward_fixed32(X32, *graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        HCSPMM.forward_fixed32(X32, *graph)
    t_host = (time.perf_counter() - t0) / 500 * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 500 * 1e6
    print("%s: HCSPMM.forward_fixed32 D=32: host issue %.1f us/call, wall %.1f us/call (500 calls, one sync)" % (name, t_host, t_all))
    for model, conv in (("gcn", GCNConv), ("gin", GINConv)):
        output = torch.zeros(n, 32, device=dev)
        net = Net(conv, d, graph, output, 32, 6).to(dev)
        opt = torch.optim.Adam(net.parameters(), lr=0.01)

        def train():
            net.train(); opt.zero_grad()
            loss = -net().gather(1, d.y.long().unsqueeze(1)).mean()  # = F.nll_loss, see HC-SpMM_main.nll_loss
            loss.backward(); opt.step()
        for _ in range(9):
            train()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            train()
        torch.cuda.synchronize()
        print("%s: %s 6 layers dim 96 hidden 32: %.2f ms/epoch (fwd+bwd+Adam)" % (name, model, (time.perf_counter() - t0) / 20 * 1e3))
