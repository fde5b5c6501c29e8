import ctypes, sys
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, os.path.join(R,'hc-spmm_amd')]
import numpy as np
from hcspmm import graphs
L=ctypes.CDLL(sys.argv[1])
TSAN = len(sys.argv) > 2 and sys.argv[2] == 'tsan'  # threaded passes only (the library holds just the LOI translation units)
vp=ctypes.c_void_p; i64=ctypes.c_int64
def P(a): return vp(a.ctypes.data) if a is not None and a.size else vp(0)
rng=np.random.default_rng(0)
n_ok=0
for trial in range(16 if TSAN else 40):
    kind=trial%5
    N=int(rng.choice([1,15,16,17,50,64,333,1000,5000,70000]))
    if TSAN and trial < 5: N = 70000  # large enough for the threaded paths (plan pool from 4096 windows, relaxed LOI from 4096 rows)
    if kind==0: rp,col=graphs.powerlaw_graph(max(N,8),max(N,8)*int(rng.integers(1,20)),seed=trial,max_degree_frac=float(rng.choice([0.02,0.9])))
    elif kind==1: rp,col=graphs.uniform_graph(max(N,4),max(N,4)*int(rng.integers(0,8))+1,seed=trial)
    elif kind==2: rp,col=graphs.planted_dense_graph_fast(max(N,32),seed=trial,dense_fraction=0.5,k_cols=int(rng.choice([4,20,41,64,130])),fill=0.5,sparse_degree=3)
    elif kind==3: rp,col=graphs.molecule_graph(max(N,50),seed=trial)
    else:
        n=max(N,1); rp,col=np.zeros(n+1,np.int32),np.zeros(0,np.int32)
    N=len(rp)-1; E=len(col); W=(N+15)//16
    for rule in (((3,) if N >= 60000 else ()) if TSAN else (0,2,4)):
        bp=np.zeros(W,np.int32); ht=np.zeros(W,np.int32); e2c=np.zeros(E,np.int32); e2r=np.zeros(E,np.int32)
        rc=L.hcspmm_preprocess_host(P(rp),P(col),i64(N),i64(E),i64(N),rule,int(rng.choice([0,1,3])),P(bp),P(e2c),P(e2r),P(ht)); assert rc==0,rc
        for force in (None,1):
            h2=ht if force is None else np.ones_like(ht)
            words=i64(0); rc=L.hcspmm_plan_words(P(rp),i64(N),i64(E),P(bp),P(h2),None,ctypes.byref(words)); assert rc==0
            plan=np.zeros(max(words.value,64),np.int32)
            rc=L.hcspmm_plan_build(P(rp),P(col),i64(N),i64(E),i64(N),P(bp),P(e2c),P(h2),None,P(plan),i64(len(plan))); assert rc==0,rc
    perm=np.zeros(N,np.int32); gs=np.zeros(max(N,1),np.int32); ng=i64(0)
    # the relaxed parallel reorder: automatic parameters with 1 and 5 threads (same permutation), tiny batches / caps, and a permutation applied in parallel
    if E < 2000000:
        perms=[]
        for prm in ((0,0,1,0),(0,0,5,0),(3,2,4,0),(1,-1,1,0)):
            pp=(ctypes.c_int32*4)(*prm); pf=np.zeros(N,np.int32)
            rc=L.hcspmm_loi_reorder_fast(P(rp),P(col),i64(N),i64(E),pp,P(pf),P(gs),ctypes.byref(ng)); assert rc==0,rc
            assert np.array_equal(np.sort(pf),np.arange(N)); perms.append(pf)
        assert np.array_equal(perms[0],perms[1])
        rp2=np.zeros(N+1,np.int32); col2=np.zeros(E,np.int32)
        rc=L.hcspmm_apply_permutation(P(rp),P(col),i64(N),i64(E),P(perms[0]),P(rp2),P(col2)); assert rc==0
        if E:
            bad=col.copy(); bad[E//2]=N
            assert L.hcspmm_loi_reorder_fast(P(rp),P(bad),i64(N),i64(E),None,P(pf),P(gs),ctypes.byref(ng))==-1
    if TSAN:
        n_ok+=1
        continue
    if E < 400000:
        for variant in (0,1,2,3):
            rc=L.hcspmm_loi_reorder_variant(P(rp),P(col),i64(N),i64(E),variant,P(perm),P(gs),ctypes.byref(ng))
            assert rc in (0,-1), rc
        rp2=np.zeros(N+1,np.int32); col2=np.zeros(E,np.int32)
        L.hcspmm_loi_reorder_variant(P(rp),P(col),i64(N),i64(E),0,P(perm),P(gs),ctypes.byref(ng))
        rc=L.hcspmm_apply_permutation(P(rp),P(col),i64(N),i64(E),P(perm),P(rp2),P(col2)); assert rc==0
    if N >= 50 and E > 0 and not TSAN:  # a row pointer that decreases, or points past the entries, in the middle of the array: refused by the window pass before any column is read
        for pos, val in ((N // 2, int(rp[N // 2 + 1]) + 5), (N // 3, E + 1000), (N // 3, -4)):
            rb = rp.copy(); rb[pos] = val
            if np.all(np.diff(rb) >= 0) and rb[pos] <= E and rb[pos] >= 0: continue
            rc=L.hcspmm_preprocess_host(P(rb),P(col),i64(N),i64(E),i64(N),3,int(rng.choice([0,1,3])),P(bp),P(e2c),P(e2r),P(ht)); assert rc==-1,rc
    if N >= 50 and E > 0:  # malformed row pointers must be refused before any column is read (all LOI variants, plan, preprocess)
        for bad0, badN in ((1, rp[-1]), (0, rp[-1] + 7), (0, max(rp[-1] - 1, 0)), (-3, rp[-1])):
            rb = rp.copy(); rb[0] = bad0; rb[-1] = badN
            for variant in (0,1,2,3):
                rc=L.hcspmm_loi_reorder_variant(P(rb),P(col),i64(N),i64(E),variant,P(perm),P(gs),ctypes.byref(ng)); assert rc==-1,(variant,rc)
            rc=L.hcspmm_plan_build(P(rb),P(col),i64(N),i64(E),i64(N),P(bp),P(e2c),P(ht),None,P(plan),i64(len(plan))); assert rc==-1,rc
    # XCD-affine column slices forced on (any size), 8 and 24 slices
    for thr,ns in ((1,8),(5,24)):
        pp=(ctypes.c_int32*5)(0,0,0,thr,ns)
        words=i64(0); rc=L.hcspmm_plan_words(P(rp),i64(N),i64(E),P(bp),P(ht),pp,ctypes.byref(words)); assert rc==0
        plan=np.zeros(max(words.value,64),np.int32)
        rc=L.hcspmm_plan_build(P(rp),P(col),i64(N),i64(E),i64(N),P(bp),P(e2c),P(ht),pp,P(plan),i64(len(plan))); assert rc==0,rc
    n_ok+=1
print(('tsan' if TSAN else 'asan/ubsan')+' host run ok:',n_ok,'graphs')
