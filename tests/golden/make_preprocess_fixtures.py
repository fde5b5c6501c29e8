"""Generates tests/golden/preprocess_ints.npz: seeded small graphs + the four integer products of
`preprocess` under every classifier rule, as computed by the ORACLE (oracle/hcspmm_oracle.c).

The reference holds no golden vectors for this path and its CUDA preprocess cannot run here
(SURVEY.md 8c), so these are NOT reference outputs: they freeze the restatement of
hybrid_all_kernel.cu:213-408 (checked against SURVEY.md Appendix A's known-answer table and hand cases
in tests/test_oracle_cpu.py) so that neither the oracle nor the product can drift unnoticed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import oracle  # noqa: E402
from hcspmm import graphs  # noqa: E402

if __name__ == "__main__":
    cases = {
        "powerlaw_777": graphs.powerlaw_graph(777, 9000, seed=21),
        "planted_640": graphs.planted_dense_graph(640, seed=22),
        "uniform_300": graphs.uniform_graph(300, 2500, seed=23),
    }
    store = {}
    for name, (rp, col) in cases.items():
        store[name + "_row_pointers"], store[name + "_column_index"] = rp, col
        for rule in (0, 1, 2, 3, 4):
            bp, e2c, e2r, ht = oracle.preprocess(rp, col, rule)
            store["%s_rule%d_blockPartition" % (name, rule)] = bp
            store["%s_rule%d_hybrid_type" % (name, rule)] = ht
        store[name + "_edgeToColumn"], store[name + "_edgeToRow"] = e2c, e2r  # rule-independent
        print(name, "N", len(rp) - 1, "E", len(col), "dense windows by rule",
              [int(store["%s_rule%d_hybrid_type" % (name, r)].sum()) for r in range(5)])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preprocess_ints.npz"), **store)
