"""ctypes binding of libhcspmm.so -- one declaration per symbol of include/hcspmm.h."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HCSPMM_LIB: another build of the same library (A/B runs of kernel variants on one box); default = the in-tree build
LIB_PATH = os.environ.get("HCSPMM_LIB") or os.path.normpath(os.path.join(_HERE, "..", "csrc", "libhcspmm.so"))

RULE_INTENDED = 0
RULE_INTENDED_GUARD = 1
RULE_AS_SHIPPED = 2
RULE_MI355X = 3
RULE_MI355X_WIDE = 4

OK, EINVAL, ENOMEM, EPLAN, EHIP, EWORKSPACE, ERANGE = 0, -1, -2, -3, -4, -5, -6


class Header(ctypes.Structure):
    """hcspmm_plan_header (include/hcspmm.h)."""
    WORDS = 64
    MAGIC = 0x48435350
    _fields_ = [(n, ctypes.c_int32) for n in (
        "magic", "version", "total_words", "num_nodes", "num_edges", "num_windows", "split_threshold", "segment_len",
        "n_tasks", "n_dense", "n_split_rows", "n_partials", "off_tasks", "off_dense_index", "off_dense_pack",
        "off_fixups", "nnz_sparse", "nnz_dense", "uniq_dense", "max_dense_k")] + [("n_len_gt", ctypes.c_int32 * 5),
                                                                                  ("n_tiny", ctypes.c_int32), ("n_dense_compact", ctypes.c_int32),
                                                                                  ("off_dense_compact", ctypes.c_int32), ("n_dense_compact2", ctypes.c_int32),
                                                                                  ("off_dense_compact2", ctypes.c_int32), ("num_columns", ctypes.c_int32),
                                                                                  ("n_sparse_windows", ctypes.c_int32), ("off_sparse_windows", ctypes.c_int32),
                                                                                  ("fingerprint_lo", ctypes.c_uint32), ("fingerprint_hi", ctypes.c_uint32),
                                                                                  ("dense_k_sum", ctypes.c_int32), ("flags", ctypes.c_int32),
                                                                                  ("n_slices", ctypes.c_int32), ("slice_threshold", ctypes.c_int32),
                                                                                  ("off_slice_table", ctypes.c_int32), ("off_slice_tasks", ctypes.c_int32),
                                                                                  ("n_slice_tasks", ctypes.c_int32), ("slice_xcd_tasks", ctypes.c_int32),
                                                                                  ("nnz_sliced", ctypes.c_int32), ("n_sliced_rows", ctypes.c_int32),
                                                                                  ("panel_cols", ctypes.c_int32), ("reserved", ctypes.c_int32 * 18)]

    @property
    def fingerprint(self):
        return (int(self.fingerprint_hi) << 32) | int(self.fingerprint_lo)


class PlanParams(ctypes.Structure):
    """hcspmm_plan_params."""
    _fields_ = [("split_threshold", ctypes.c_int32), ("segment_len", ctypes.c_int32), ("fuse_in_launch", ctypes.c_int32),
                ("slice_threshold", ctypes.c_int32), ("n_slices", ctypes.c_int32), ("panel_cols", ctypes.c_int32)]


# every exported symbol of include/hcspmm.h: name -> (restype, argtypes)
_vp, _i64, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
_hp, _pp = ctypes.POINTER(Header), ctypes.POINTER(PlanParams)
SYMBOLS = {
    "hcspmm_strerror": (ctypes.c_char_p, [_int]),
    "hcspmm_abi_version": (_int, []),
    "hcspmm_last_hip_error": (_int, []),
    "hcspmm_preprocess_host": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "hcspmm_edge_to_row_device": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "hcspmm_plan_words": (_int, [_vp, _i64, _i64, _vp, _vp, _pp, ctypes.POINTER(_i64)]),
    "hcspmm_plan_build": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _pp, _vp, _i64]),
    "hcspmm_plan_check": (_int, [_hp, _i64, _i64, _i64]),
    "hcspmm_graph_fingerprint_host": (_int, [_vp, _vp, _i64, _i64, ctypes.POINTER(ctypes.c_uint64)]),
    "hcspmm_graph_fingerprint_device": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "hcspmm_workspace_bytes": (_sz, [_hp, _int]),
    "hcspmm_wide_threshold": (ctypes.c_int32, [_hp, _int]),
    "hcspmm_wide_threshold_typed": (ctypes.c_int32, [_hp, _int, _int]),
    "hcspmm_forward_typed": (_int, [_vp, _i64, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _hp, _i64, _i64, _int, _vp,
                                    _sz, _vp]),
    "hcspmm_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _hp, _i64, _i64, _int, _vp, _sz, _vp]),
    "hcspmm_forward_strided": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _hp, _i64, _i64, _int, _vp, _sz,
                                      _vp]),
    "hcspmm_forward_fused": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _hp,
                                    _i64, _i64, _int, _vp, _sz, _vp]),
    "hcspmm_fused_in_launch": (_int, [_hp, _int, _int]),
    "hcspmm_dense_update": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _int, _int, _vp]),
    "hcspmm_weight_grad_workspace": (_sz, [_i64, _int, _int]),
    "hcspmm_weight_grad": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _int, _vp, _sz, _vp]),
    "hcspmm_loi_reorder": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, ctypes.POINTER(_i64)]),
    "hcspmm_loi_reorder_variant": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, ctypes.POINTER(_i64)]),
    "hcspmm_loi_reorder_fast": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, ctypes.POINTER(_i64)]),
    "hcspmm_apply_permutation": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
}

_LIB = None


def lib():
    """Load libhcspmm.so (built in-tree by hc-spmm_amd/csrc/Makefile).  Fails loudly when absent:
    there is no CPU or PyTorch fallback for the hot path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libhcspmm.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (or make -C hc-spmm_amd/csrc); no fallback path exists" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, soft=False):
    if rc != 0 and not soft:
        L = lib()
        msg = L.hcspmm_strerror(rc).decode()
        if rc == EHIP:
            msg += " (hipError_t %d)" % L.hcspmm_last_hip_error()
        raise RuntimeError("hcspmm: %s [code %d]" % (msg, rc))
    return rc
