import sys, time
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT,os.path.join(ROOT,'hc-spmm_amd')]
import numpy as np, torch
import hcspmm
from hcspmm import graphs
dev=torch.device('cuda:0')
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for name,(rp,col) in (('cora',graphs.powerlaw_graph(10000,50000,seed=1)),('reddit',graphs.powerlaw_graph(233000,11600000,seed=3))):
    N,E=len(rp)-1,len(col)
    rp_d,col_d=torch.from_numpy(rp).to(dev),torch.from_numpy(col).to(dev)
    t0=time.perf_counter(); outs=hcspmm.preprocess(col_d,rp_d,N,E,(N+15)//16); torch.cuda.synchronize(); print(name,'preprocess %.1f ms'%((time.perf_counter()-t0)*1e3))
    t0=time.perf_counter(); outs=hcspmm.preprocess(col_d,rp_d,N,E,(N+15)//16); torch.cuda.synchronize(); print(name,'preprocess(2nd) %.1f ms'%((time.perf_counter()-t0)*1e3))
    for D,H in ((32,32),(128,32),(64,64)):
        X=torch.randn(N,D,device=dev); W=torch.randn(D,H,device=dev)
        a=(rp_d,col_d,*outs)
        t_sp=timeit(lambda: hcspmm.forward(X,*a))
        t_fu=timeit(lambda: hcspmm.forward_fixed32_fused(X,*a,W))
        Z=hcspmm.forward(X,*a)[0]
        t_mm=timeit(lambda: torch.mm(Z,W))
        print(name,'D',D,'H',H,'spmm %.1f us  fused(spmm+update) %.1f us  torch.mm alone %.1f us'%(t_sp,t_fu,t_mm))
