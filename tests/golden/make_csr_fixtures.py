"""Generates tests/golden/csr_from_text.npz -- run ONLY in the authoring container.

Feeds small COO text files (our own synthetic inputs: "dst,src" 1-based lines, with duplicate
edges, unsorted lines, isolated ids) to the REFERENCE's loader, `HCSPMM_dataset.init_edges`
(/root/reference/dataset.py:43-103), imported from /root/reference, and records the CSR it
builds (row_pointers / column_index / num_nodes / num_edges).  The loader's last statement moves
a tensor to the GPU (dataset.py:107) and raises "No HIP GPUs are available" here -- an ordinary
Python error after the CSR attributes are already set -- which is caught.  Only inputs and
expected outputs are stored; the loader itself does not travel.
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csr_from_text.npz")


def ref_init_edges(text):
    sys.path.insert(0, REF)
    import dataset as ref_dataset  # the reference's dataset.py
    obj = ref_dataset.HCSPMM_dataset.__new__(ref_dataset.HCSPMM_dataset)
    import torch
    torch.nn.Module.__init__(obj)
    obj.nodes = set()
    obj.load_from_txt = True
    obj.verbose_flag = False
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        obj.init_edges(path)
    except (RuntimeError, AssertionError) as e:  # .cuda() without a GPU (dataset.py:107)
        assert "HIP" in str(e) or "CUDA" in str(e) or "cuda" in str(e), e
    finally:
        os.unlink(path)
    return (obj.row_pointers.numpy().astype(np.int32), obj.column_index.numpy().astype(np.int32),
            int(obj.num_nodes), int(obj.num_edges))


def make_text(seed, n, m, dup=0.2):
    rng = np.random.default_rng(seed)
    src = rng.integers(1, n + 1, m)
    dst = rng.integers(1, n + 1, m)
    k = int(m * dup)
    src = np.concatenate([src, src[:k]])
    dst = np.concatenate([dst, dst[:k]])
    p = rng.permutation(len(src))
    return "".join("%d,%d\n" % (d, s) for d, s in zip(dst[p], src[p]))


if __name__ == "__main__":
    store = {}
    texts = {
        "five_nodes": "2,1\n1,2\n3,2\n2,3\n5,4\n4,5\n2,1\n",
        "rand_40": make_text(1, 40, 150),
        "rand_333": make_text(2, 333, 2500),
    }
    for name, text in texts.items():
        rp, col, n, e = ref_init_edges(text)
        store[name + "_text"] = np.frombuffer(text.encode(), np.uint8)
        store[name + "_row_pointers"] = rp
        store[name + "_column_index"] = col
        store[name + "_num_nodes"] = np.int64(n)
        store[name + "_num_edges"] = np.int64(e)
        print(name, "N", n, "raw edges", e, "nnz", len(col))
    np.savez_compressed(OUT, **store)
