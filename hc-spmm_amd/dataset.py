"""Graph loading for the GNN driver -- counterpart of the reference's dataset.py (class
HCSPMM_dataset, dataset.py:8-121), same constructor and attributes.

On-disk formats: text = one "dst,src" line per edge, 1-based ids
(dataset.py:52-53: row = src = 2nd field, column = dst = 1st field); .npz = arrays `src_li`,
`dst_li`, scalar `num_nodes` (dataset.py:69-79).  The CSR is binary: duplicate edges merge and
columns come out ascending per row, as scipy's coo->csr does for the reference (dataset.py:95-96).
tests/golden/csr_from_text.npz holds CSRs produced by the reference loader for the same text.
"""
import time

import numpy as np
import scipy.sparse as sp
import torch

from config import func


def _default_device():
    return torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


class HCSPMM_dataset(torch.nn.Module):
    def __init__(self, path, dim, num_class, load_from_txt=True, verbose=False, device=None, seed=None):
        super().__init__()
        self.device = torch.device(device) if device is not None else _default_device()
        self.load_from_txt = load_from_txt
        self.verbose_flag = verbose
        self.reorder_flag = False
        self.num_features = dim
        self.num_classes = num_class
        self.nodes = set()
        self.init_edges(path)
        self.init_embedding(dim, seed)
        self.init_labels(num_class)
        n = self.num_nodes
        # the reference builds 100 % / 30 % / 10 % prefix masks (dataset.py:33-41); train() ignores them
        for name, frac in (("train_mask", 1.0), ("val_mask", 0.3), ("test_mask", 0.1)):
            m = torch.zeros(n, dtype=torch.bool)
            m[:int(n * frac)] = True
            setattr(self, name, m.to(self.device))

    # -- edges ---------------------------------------------------------------------------------
    def _read_edges(self, path):
        if self.load_from_txt:
            raw = np.loadtxt(path, delimiter=",", dtype=np.int64, ndmin=2)
            dst, src = raw[:, 0] - 1, raw[:, 1] - 1
            num_nodes = int(max(dst.max(), src.max())) + 1 if raw.size else 0
        else:
            if not path.endswith(".npz"):
                raise ValueError("graph file must be a .npz file")
            g = np.load(path)
            src, dst = np.asarray(g["src_li"], np.int64), np.asarray(g["dst_li"], np.int64)
            num_nodes = int(g["num_nodes"])
        return src, dst, num_nodes

    def init_edges(self, path):
        t0 = time.perf_counter()
        src, dst, self.num_nodes = self._read_edges(path)
        self.num_edges = int(src.shape[0])  # raw line count, as the reference reports it
        self.edge_index = np.stack([src, dst])
        self.avg_degree = self.num_edges / max(self.num_nodes, 1)
        self.avg_edgeSpan = float(np.mean(np.abs(src - dst))) if self.num_edges else 0.0
        csr = sp.coo_matrix((np.ones(self.num_edges, np.int8), (src, dst)),
                            shape=(self.num_nodes, self.num_nodes)).tocsr()
        csr.sum_duplicates()
        csr.sort_indices()
        self.column_index = torch.from_numpy(csr.indices.astype(np.int32))
        self.row_pointers = torch.from_numpy(csr.indptr.astype(np.int32))
        deg = np.diff(csr.indptr).astype(np.float32)
        self.degrees = torch.sqrt(torch.from_numpy(np.where(deg > 0, deg, func(0)).astype(np.float32))).to(self.device)
        if self.verbose_flag:
            print("# Loading + CSR (s): {:.3f}".format(time.perf_counter() - t0))
            print("# nodes: {}".format(self.num_nodes))
            print("# avg_degree: {:.2f}".format(self.avg_degree))
            print("# avg_edgeSpan: {}".format(int(self.avg_edgeSpan)))

    # -- features / labels (random features, all-ones labels: dataset.py:109-121) ----------------
    def init_embedding(self, dim, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        self.x = torch.randn(self.num_nodes, dim).to(self.device)

    def init_labels(self, num_class):
        self.y = torch.ones(self.num_nodes).long().to(self.device)
