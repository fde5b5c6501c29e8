"""Where a WARM hcspmm.preprocess spends its time: the phases of hcspmm/__init__.py preprocess() re-enacted with pinned buffers and a
timer around each, third call of the process.   python tools/prep_warm_breakdown.py [workload ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import numpy as np, torch
import bench, hcspmm
from hcspmm import capi

dev = torch.device("cuda:0")
L = capi.lib()
for wl in sys.argv[1:] or ["reddit", "rd_like"]:
    n_local, e_local, _, vw, _ = bench.WORKLOADS[wl]
    rp, col = bench.make_local_block(wl, n_local, e_local, vw, 0)
    col_d, rp_d = torch.from_numpy(col).to(dev), torch.from_numpy(rp).to(dev)
    N, E = len(rp) - 1, len(col)
    W, M = (N + 15) // 16, n_local * vw
    for rep in range(3):
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        def lap(sync=False):
            if sync:
                torch.cuda.synchronize()
            t.append(time.perf_counter())
        col_h = torch.empty(E, dtype=torch.int32, pin_memory=True); rp_h = torch.empty(N + 1, dtype=torch.int32, pin_memory=True); lap()
        col_h.copy_(col_d, non_blocking=True); rp_h.copy_(rp_d, non_blocking=True); lap(True)
        bp = torch.empty(W, dtype=torch.int32, pin_memory=True); ht = torch.empty(W, dtype=torch.int32, pin_memory=True)
        e2c = torch.empty(E, dtype=torch.int32, pin_memory=True); e2r = torch.empty(E, dtype=torch.int32, device=dev); lap()
        L.hcspmm_edge_to_row_device(rp_d.data_ptr(), N, E, e2r.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); lap()
        L.hcspmm_preprocess_host(rp_h.data_ptr(), col_h.data_ptr(), N, E, M, 3, 0, bp.data_ptr(), e2c.data_ptr(), None, ht.data_ptr()); lap()
        words = ctypes.c_int64(0)
        L.hcspmm_plan_words(rp_h.data_ptr(), N, E, bp.data_ptr(), ht.data_ptr(), None, ctypes.byref(words)); lap()
        plan = torch.empty(words.value, dtype=torch.int32, pin_memory=True); lap()
        L.hcspmm_plan_build(rp_h.data_ptr(), col_h.data_ptr(), N, E, M, bp.data_ptr(), e2c.data_ptr(), ht.data_ptr(), None, plan.data_ptr(), plan.numel()); lap()
        outs = [x.to(dev, non_blocking=True) for x in (bp, e2c, ht, plan)]; lap(True)
        d = np.diff(t) * 1e3
        if rep == 2:
            print("%s warm: pinned allocs (inputs) %.2f | D2H %.2f | pinned / device allocs (outputs) %.2f | edgeToRow launch %.2f | window pass %.2f | plan words %.2f | "
                  "plan alloc %.2f | plan build %.2f | H2D %.2f | total %.2f ms (E = %d, plan %d words)" % (wl, *d, d.sum(), E, words.value))
        del col_h, rp_h, bp, ht, e2c, e2r, plan, outs
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); o = hcspmm.preprocess(col_d, rp_d, N, E, W, num_columns=M); torch.cuda.synchronize()
        print(wl, "hcspmm.preprocess call %d: %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
        del o
