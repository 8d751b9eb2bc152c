"""TEST INFRASTRUCTURE ONLY: ctypes access to the CPU restatement (oracle/libhm_oracle.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhm_oracle.so")


class Cfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("qp", C.c_int), ("wpp", C.c_int),
                ("lambda_", C.c_double), ("chroma_weight", C.c_double)]


CTU_DTYPE = np.dtype([("total_cost", "<f8"), ("total_bits", "<u4"), ("total_dist", "<u4"),
                      ("depth", "u1", 256), ("part_size", "u1", 256), ("pred_mode", "u1", 256),
                      ("intra_dir_luma", "u1", 256), ("intra_dir_chroma", "u1", 256), ("tr_idx", "u1", 256),
                      ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
                      ("coeff_y", "<i4", 4096), ("coeff_cb", "<i4", 1024), ("coeff_cr", "<i4", 1024)])

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.hmo_cfg_set_qp.argtypes = [C.POINTER(Cfg), C.c_int]
        _lib.hmo_compress_rows.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for n in ("hmo_sad", "hmo_sse", "hmo_hads"):
            getattr(_lib, n).restype = C.c_uint32
    return _lib


def compress(planes, bit_depth, qp, wpp=0, max_ctus=0):
    """planes: (Y,U,V) uint16 -> (rec planes, ctus structured array)"""
    L = lib()
    h, w = planes[0].shape
    cfg = Cfg(w, h, bit_depth, qp, wpp, 0.0, 0.0)
    L.hmo_cfg_set_qp(C.byref(cfg), qp)
    n = ((w + 63) // 64) * ((h + 63) // 64)
    org = [np.ascontiguousarray(p, np.uint16) for p in planes]
    rec = [np.zeros_like(p) for p in org]
    ctus = np.zeros(n, CTU_DTYPE)
    po = (C.c_void_p * 3)(*[p.ctypes.data for p in org])
    pr = (C.c_void_p * 3)(*[p.ctypes.data for p in rec])
    rc = L.hmo_compress_rows(C.byref(cfg), po, pr, ctus.ctypes.data, max_ctus)
    if rc != 0:
        raise RuntimeError(f"oracle failed rc={rc}")
    return rec, ctus


def dist(kind, org, cur, bit_depth, sub_shift=0):
    L = lib()
    n = org.shape[0]
    o, c = np.ascontiguousarray(org, np.int16), np.ascontiguousarray(cur, np.int16)
    if kind == 0:
        return L.hmo_sad(o.ctypes.data_as(C.c_void_p), n, c.ctypes.data_as(C.c_void_p), n, n, n, sub_shift, bit_depth)
    if kind == 1:
        return L.hmo_sse(o.ctypes.data_as(C.c_void_p), n, c.ctypes.data_as(C.c_void_p), n, n, n, bit_depth)
    return L.hmo_hads(o.ctypes.data_as(C.c_void_p), n, c.ctypes.data_as(C.c_void_p), n, n, n, bit_depth)


def transform(inverse, block, bit_depth, use_dst=0):
    L = lib()
    n = block.shape[0]
    b = np.ascontiguousarray(block, np.int32)
    out = np.zeros_like(b)
    f = L.hmo_inv_transform if inverse else L.hmo_fwd_transform
    f(bit_depth, b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n, use_dst)
    return out
