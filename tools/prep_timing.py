import sys, time, ctypes
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT,os.path.join(ROOT,'hc-spmm_amd')]
import numpy as np, torch
import hcspmm
from hcspmm import graphs, capi
L=capi.lib()
dev=torch.device('cuda:0')
for name,(rp,col) in (('reddit',graphs.powerlaw_graph(233000,11600000,seed=3)),('rd_like',graphs.powerlaw_graph(4859280,10149830,seed=3)),('dense',graphs.planted_dense_graph_fast(2000000,seed=3,dense_fraction=0.7,k_cols=20,fill=0.45,sparse_degree=16))):
    N,E=len(rp)-1,len(col); W=(N+15)//16
    col_d,rp_d=torch.from_numpy(col).to(dev),torch.from_numpy(rp).to(dev)
    torch.cuda.synchronize()
    for rep in range(2):
        t0=time.perf_counter(); col_h=col_d.cpu(); rp_h=rp_d.cpu(); t1=time.perf_counter()
        bp=torch.zeros(W,dtype=torch.int32); ht=torch.zeros(W,dtype=torch.int32); e2c=torch.zeros(E,dtype=torch.int32); e2r=torch.zeros(E,dtype=torch.int32)
        t2=time.perf_counter()
        for th in (0,16,32,64):
            ta=time.perf_counter()
            L.hcspmm_preprocess_host(rp_h.data_ptr(),col_h.data_ptr(),N,E,N,0,th,bp.data_ptr(),e2c.data_ptr(),e2r.data_ptr(),ht.data_ptr())
            tb=time.perf_counter()
            if rep: print(name,'preprocess_host threads',th,'%.1f ms'%((tb-ta)*1e3))
        t3=time.perf_counter()
        words=ctypes.c_int64(0); L.hcspmm_plan_words(rp_h.data_ptr(),N,E,bp.data_ptr(),ht.data_ptr(),None,ctypes.byref(words))
        plan=torch.zeros(words.value,dtype=torch.int32)
        t4=time.perf_counter()
        L.hcspmm_plan_build(rp_h.data_ptr(),col_h.data_ptr(),N,E,N,bp.data_ptr(),e2c.data_ptr(),ht.data_ptr(),None,plan.data_ptr(),plan.numel())
        t5=time.perf_counter()
        outs=[t.to(dev) for t in (bp,e2c,e2r,ht,plan)]; torch.cuda.synchronize(); t6=time.perf_counter()
        if rep: print(name,'D2H %.1f alloc %.1f planwords+alloc %.1f plan_build %.1f H2D %.1f ms (E=%d, plan %d words)'%((t1-t0)*1e3,(t2-t1)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t6-t5)*1e3,E,words.value))
    for rep in range(3):
        t0=time.perf_counter(); o=hcspmm.preprocess(col_d,rp_d,N,E,W); torch.cuda.synchronize(); print(name,'total hcspmm.preprocess %.1f ms'%((time.perf_counter()-t0)*1e3))
