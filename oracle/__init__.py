"""oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (see hcspmm_oracle.c / loi_oracle.py
for the reference file:line each function follows).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product in hc-spmm_amd/ never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RULE_INTENDED = 0        # paper p.7 / hybrid_all_kernel.cu:261 without the size>32 guard
RULE_INTENDED_GUARD = 1  # hybrid_all_kernel.cu:261 literally
RULE_AS_SHIPPED = 2      # hybrid_all_kernel.cu:262 literally (float used as bool)
RULE_MI355X = 3          # not a reference rule: the product's MI355X refit of the same two-feature logit (narrow embeddings)
RULE_MI355X_WIDE = 4     # ditto, fit at embedding width 128


def build(force=False):
    """Compile liboracle.so with gcc (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hcspmm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/LOI.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        L.hcspmm_oracle_logit.restype = ctypes.c_double
        L.hcspmm_oracle_logit.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_int32]
        L.hcspmm_oracle_classify.restype = ctypes.c_int32
        L.hcspmm_oracle_classify.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int]
        L.hcspmm_oracle_preprocess.argtypes = [i32p, i32p, i64, i64, ctypes.c_int, i32p, i32p, i32p, i32p]
        L.hcspmm_oracle_spmm_f32.argtypes = [i32p, i32p, i64, i64, f32p, f32p]
        L.hcspmm_oracle_spmm_f64.argtypes = [i32p, i32p, i64, i64, f32p, f64p]
        L.hcspmm_oracle_spmm_abs_f64.argtypes = [i32p, i32p, i64, i64, f32p, f64p]
        L.hcspmm_oracle_spmm_hybrid_f32.argtypes = [i32p, i32p, i32p, i32p, i32p, i32p, i64, i64, f32p, f32p]
        L.hcspmm_oracle_spmm_fused_f32.argtypes = [i32p, i32p, i64, i64, i64, f32p, f32p, f32p, f32p]
        _LIB = L
    return _LIB


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def logit(size, nnz_window, num):
    return lib().hcspmm_oracle_logit(int(size), int(nnz_window), int(num))


def classify(size, nnz_window, num, rule=RULE_INTENDED):
    return lib().hcspmm_oracle_classify(int(size), int(nnz_window), int(num), int(rule))


def preprocess(rowptr, col, rule=RULE_INTENDED):
    """-> (blockPartition[W], edgeToColumn[E], edgeToRow[E], hybrid_type[W]) int32."""
    rowptr, prp = _i32(rowptr)
    col, pc = _i32(col)
    N, E = rowptr.shape[0] - 1, col.shape[0]
    W = (N + 15) // 16
    bp = np.zeros(W, np.int32)
    e2c = np.zeros(E, np.int32)
    e2r = np.zeros(E, np.int32)
    ht = np.zeros(W, np.int32)
    rc = lib().hcspmm_oracle_preprocess(prp, pc, N, E, rule, _i32(bp)[1], _i32(e2c)[1], _i32(e2r)[1], _i32(ht)[1])
    if rc != 0:
        raise RuntimeError("oracle preprocess failed rc=%d" % rc)
    return bp, e2c, e2r, ht


def spmm_f32(rowptr, col, X):
    rowptr, prp = _i32(rowptr)
    col, pc = _i32(col)
    X, px = _f32(X)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Z = np.empty((N, D), np.float32)
    lib().hcspmm_oracle_spmm_f32(prp, pc, N, D, px, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return Z


def spmm_f64(rowptr, col, X, absolute=False):
    rowptr, prp = _i32(rowptr)
    col, pc = _i32(col)
    X, px = _f32(X)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Z = np.empty((N, D), np.float64)
    fn = lib().hcspmm_oracle_spmm_abs_f64 if absolute else lib().hcspmm_oracle_spmm_f64
    fn(prp, pc, N, D, px, Z.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return Z


def spmm_hybrid_f32(rowptr, col, bp, e2c, e2r, ht, X):
    rowptr, prp = _i32(rowptr)
    col, pc = _i32(col)
    X, px = _f32(X)
    N, D = rowptr.shape[0] - 1, X.shape[1]
    Z = np.empty((N, D), np.float32)
    rc = lib().hcspmm_oracle_spmm_hybrid_f32(prp, pc, _i32(bp)[1], _i32(e2c)[1], _i32(e2r)[1], _i32(ht)[1], N, D, px,
                                             Z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc != 0:
        raise RuntimeError("oracle hybrid spmm failed rc=%d" % rc)
    return Z


def spmm_fused_f32(rowptr, col, X, Wt):
    """-> (out = (A X) Wt  [N,H], out2 = A X [N,D])."""
    rowptr, prp = _i32(rowptr)
    col, pc = _i32(col)
    X, px = _f32(X)
    Wt, pw = _f32(Wt)
    N, D, H = rowptr.shape[0] - 1, X.shape[1], Wt.shape[1]
    assert Wt.shape[0] == D
    out = np.empty((N, H), np.float32)
    out2 = np.empty((N, D), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib().hcspmm_oracle_spmm_fused_f32(prp, pc, N, D, H, px, pw, out.ctypes.data_as(fp), out2.ctypes.data_as(fp))
    return out, out2


def check_spmm(got, rowptr, col, X, rel=1e-5):
    """Parity criterion for floating-point A*X (north_star: 1e-5 relative fp32).

    |got - Z64| <= rel * sum_j |X[col_j, d]|  componentwise, where Z64 is the fp64-accumulated
    product: the summation's own scale, so entries whose terms cancel are judged fairly.  Returns
    (ok, max_ratio) with max_ratio = max |err| / (rel * scale).
    """
    z64 = spmm_f64(rowptr, col, X)
    zabs = spmm_f64(rowptr, col, X, absolute=True)
    err = np.abs(np.asarray(got, np.float64) - z64)
    tiny = np.finfo(np.float32).tiny
    ratio = err / (rel * zabs + tiny)
    m = float(ratio.max()) if ratio.size else 0.0
    return m <= 1.0, m
