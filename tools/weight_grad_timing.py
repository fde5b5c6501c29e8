"""dW = X^T dY with K = N rows (233 K) and a tiny output (D x H): the op that dominates a GCN / GIN epoch on this
box (profiles/r01/gnn_epoch_kernels.log).  Times torch.mm against re-formulations of the same product."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hc-spmm_amd")]
import hcspmm
dev = torch.device("cuda:0")


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


N = 233000
for D, H in ((96, 32), (32, 32), (32, 22), (128, 64)):
    A, B = torch.randn(N, D, device=dev), torch.randn(N, H, device=dev)
    ref = torch.mm(A.t().double(), B.double())
    res = {}
    res["mm(A.t(), B)"] = (lambda: torch.mm(A.t(), B))
    res["mm(B.t(), A).t()"] = (lambda: torch.mm(B.t(), A).t())
    res["hcspmm.weight_grad (native)"] = (lambda: hcspmm.weight_grad(A, B))
    for G in (64, 256, 1024):
        n = (N // G) * G

        def splitk(G=G, n=n):
            out = torch.bmm(A[:n].view(G, n // G, D).transpose(1, 2), B[:n].view(G, n // G, H)).sum(0)
            if n < N:
                out = out + torch.mm(A[n:].t(), B[n:])
            return out
        res["split-K bmm G=%d + sum" % G] = splitk
    print("D=%d H=%d  (ideal at 6 TB/s: %.0f us)" % (D, H, N * (D + H) * 4 / 6e12 * 1e6))
    for k, fn in res.items():
        if fn() is None:
            print("   %-28s (shape outside the kernel's range)" % k)
            continue
        err = float((fn().double() - ref).abs().max() / ref.abs().max())
        print("   %-28s %8.1f us   rel err %.1e" % (k, t_us(fn), err))
