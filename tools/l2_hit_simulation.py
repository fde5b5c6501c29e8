"""LRU model of the per-XCD L2 on the Reddit-scale headline (one 32-column panel = one 128-byte line per X row): hit rate of the
gathers under different task orders and cache capacities.  Host-only (numpy); results quoted in DESIGN.md section 8.
  python tools/l2_hit_simulation.py"""
import sys,time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,'hc-spmm_amd')]
import numpy as np
from collections import OrderedDict
from hcspmm import graphs
rp,col=graphs.powerlaw_graph(233000,11600000,seed=3)
N=len(rp)-1; deg=np.diff(rp)
indeg=np.bincount(col,minlength=N)
def length_class(l):
    c=np.zeros_like(l); x=l.copy()
    c[l>0]=np.ceil(np.log2(np.maximum(l,1)))[l>0].astype(int)+1
    return c
cls=length_class(deg)
def order_current():
    return np.lexsort((np.arange(N),-cls))   # class desc, row asc
def order_hub():
    # within class by the row's most popular column
    top=np.zeros(N,np.int64)
    for r in range(N):
        seg=col[rp[r]:rp[r+1]]
        top[r]=seg[np.argmax(indeg[seg])] if len(seg) else -1
    return np.lexsort((top,-cls))
def order_random():
    return np.random.default_rng(0).permutation(N)
def simulate(order, lines=32768, xcds=8, group=8):
    # tasks dealt to XCDs round-robin in groups of `group` rows (a workgroup = 8 tasks at D=128 panel-major L=8? keep simple)
    hits=0; tot=0
    caches=[OrderedDict() for _ in range(xcds)]
    for gi in range(0,N,group):
        x=(gi//group)%xcds; c=caches[x]
        for r in order[gi:gi+group]:
            for cc in col[rp[r]:rp[r+1]]:
                tot+=1
                if cc in c:
                    hits+=1; c.move_to_end(cc)
                else:
                    c[cc]=1
                    if len(c)>lines: c.popitem(last=False)
    return hits/tot
for name,fn in (('current',order_current),('random',order_random),('hub-sorted',order_hub)):
    t=time.time(); o=fn(); h=simulate(o); print(name,'L2 hit %.3f'%h,'(%.0fs)'%(time.time()-t),flush=True)
o=order_current()
for lines,x in ((262144,1),(65536,8),(16384,8)):
    print('lines per cache',lines,'caches',x,'hit %.3f'%simulate(o,lines=lines,xcds=x),flush=True)
