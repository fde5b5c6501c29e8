"""Where a COLD (first in the process) hcspmm.preprocess spends its time: the phases of hcspmm/__init__.py preprocess()
re-enacted one by one with a timer around each, first call and second call.   python tools/prep_cold_breakdown.py [workload]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np
import torch
import bench
import hcspmm
from hcspmm import capi

wl = sys.argv[1] if len(sys.argv) > 1 else "reddit"
n_local, e_local, _, vw, _ = bench.WORKLOADS[wl]
rp, col = bench.make_local_block(wl, n_local, e_local, vw, 0)
dev = torch.device("cuda:0")
col_d, rp_d = torch.from_numpy(col).to(dev), torch.from_numpy(rp).to(dev)
torch.cuda.synchronize()
L = capi.lib()
N, E = len(rp) - 1, len(col)
W = (N + 15) // 16
M = n_local * vw
for rep in ("cold", "warm"):
    t = [time.perf_counter()]
    def lap():
        torch.cuda.synchronize()
        t.append(time.perf_counter())
    col_h = hcspmm._i32_host(col_d); rp_h = hcspmm._i32_host(rp_d); lap()
    bp = torch.empty(W, dtype=torch.int32); ht = torch.empty(W, dtype=torch.int32); e2c = torch.empty(E, dtype=torch.int32); lap()
    L.hcspmm_preprocess_host(col_h.data_ptr() * 0 + rp_h.data_ptr(), col_h.data_ptr(), N, E, M, 0, 0, bp.data_ptr(), e2c.data_ptr(), None, ht.data_ptr()); lap()
    words = ctypes.c_int64(0)
    L.hcspmm_plan_words(rp_h.data_ptr(), N, E, bp.data_ptr(), ht.data_ptr(), None, ctypes.byref(words))
    plan = torch.empty(words.value, dtype=torch.int32); lap()
    L.hcspmm_plan_build(rp_h.data_ptr(), col_h.data_ptr(), N, E, M, bp.data_ptr(), e2c.data_ptr(), ht.data_ptr(), None, plan.data_ptr(), plan.numel()); lap()
    rp64 = rp_d.to(torch.int64)
    e2r = torch.repeat_interleave(torch.arange(N, dtype=torch.int32, device=dev), rp64[1:] - rp64[:-1], output_size=E); lap()
    outs = [x.to(dev) for x in (bp, e2c, ht, plan)]; lap()
    d = np.diff(t) * 1e3
    print("%s %s: D2H(pinned) %.1f | host allocs %.1f | window pass %.1f | plan words+alloc %.1f | plan build %.1f | edgeToRow on device %.1f | H2D %.1f | total %.1f ms"
          % (wl, rep, *d, d.sum()))
for rep in range(3):
    t0 = time.perf_counter(); o = hcspmm.preprocess(col_d, rp_d, N, E, W, num_columns=M); torch.cuda.synchronize()
    print(wl, "hcspmm.preprocess call %d: %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3))
