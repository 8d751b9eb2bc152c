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


# ---- P slices -------------------------------------------------------------------------------------------------
class RefPic(C.Structure):
    _fields_ = [("poc", C.c_int), ("slice_type", C.c_int), ("long_term", C.c_int), ("plane", C.c_void_p * 3),
                ("pred_mode", C.c_void_p), ("mv", C.c_void_p * 2), ("ref_idx", C.c_void_p * 2),
                ("num_ref", C.c_int * 2), ("ref_poc", (C.c_int * 16) * 2), ("ref_lt", (C.c_int * 16) * 2)]


class InterSlice(C.Structure):
    _fields_ = [("slice_type", C.c_int), ("poc", C.c_int), ("cabac_init_type", C.c_int), ("num_ref_idx", C.c_int * 2),
                ("ref", (C.POINTER(RefPic) * 16) * 2),
                ("col_from_l0", C.c_int), ("col_ref_idx", C.c_int), ("tmvp", C.c_int), ("mvd_l1_zero", C.c_int),
                ("max_merge_cand", C.c_int), ("check_ldc", C.c_int),
                ("lambda_motion_sad", C.c_uint32), ("lambda_motion_sse", C.c_uint32)]


CTU_INTER_DTYPE = np.dtype([("skip", "u1", 256), ("merge_flag", "u1", 256), ("merge_idx", "u1", 256), ("inter_dir", "u1", 256),
                            ("mv", "<i2", (2, 256, 2)), ("mvd", "<i2", (2, 256, 2)),
                            ("ref_idx", "i1", (2, 256)), ("mvp_idx", "i1", (2, 256)), ("mvp_num", "i1", (2, 256))])


def _inter_slice(srec, finals):
    """(InterSlice struct, objects to keep alive) from an 'S' record of tests/hmd2.py and the 'F' records of the pictures it references"""
    keep, refs = [], {}
    for poc in set(int(srec["ref_poc"][l][i]) for l in range(2) for i in range(srec["num_ref_idx"][l])):
        f = finals[poc]
        pl = [np.ascontiguousarray(p, np.uint16) for p in f["rec"]]
        mot = f["motion"]
        pm = np.ascontiguousarray(mot["pred_mode"]); mv = [np.ascontiguousarray(mot["mv0"]), np.ascontiguousarray(mot["mv1"])]
        ri = [np.ascontiguousarray(mot["ref_idx0"]), np.ascontiguousarray(mot["ref_idx1"])]
        keep += pl + [pm] + mv + ri
        r = RefPic()
        r.poc, r.slice_type, r.long_term = poc, int(f["slice_type"]), 0
        for c in range(3):
            r.plane[c] = pl[c].ctypes.data
        r.pred_mode = pm.ctypes.data
        for l in range(2):
            r.mv[l] = mv[l].ctypes.data; r.ref_idx[l] = ri[l].ctypes.data; r.num_ref[l] = int(f["num_ref_idx"][l])
            for i in range(16):
                r.ref_poc[l][i] = int(f["ref_poc"][l][i]); r.ref_lt[l][i] = int(f["ref_long_term"][l][i])
        refs[poc] = r
    s = InterSlice()
    s.slice_type, s.poc, s.cabac_init_type = int(srec["slice_type"]), int(srec["poc"]), int(srec["cabac_init_type"])
    for l in range(2):
        s.num_ref_idx[l] = int(srec["num_ref_idx"][l])
        for i in range(s.num_ref_idx[l]):
            s.ref[l][i] = C.pointer(refs[int(srec["ref_poc"][l][i])])
    s.col_from_l0, s.col_ref_idx, s.tmvp = int(srec["col_from_l0"]), int(srec["col_ref_idx"]), int(srec["tmvp"])
    s.mvd_l1_zero, s.max_merge_cand, s.check_ldc = int(srec["mvd_l1_zero"]), int(srec["max_merge_cand"]), int(srec["check_ldc"])
    s.lambda_motion_sad, s.lambda_motion_sse = int(srec["lambda_motion_sad"]), int(srec["lambda_motion_sse"])
    keep.append(refs)
    return s, keep


def compress_inter(planes, bit_depth, srec, finals, trace=None, wpp=0):
    """One P slice.  srec: an 'S' record of tests/hmd2.py (slice parameters as the reference used them);
    finals: {poc: 'F' record} of the pictures it references.  Returns (rec planes, ctus, inter ctus)."""
    L = lib()
    h, w = planes[0].shape
    cfg = Cfg(w, h, bit_depth, int(srec["qp"]), wpp, float(srec["lambda"]), float(srec["weight_cb"]))
    n = ((w + 63) // 64) * ((h + 63) // 64)
    s, keep = _inter_slice(srec, finals)
    org = [np.ascontiguousarray(p, np.uint16) for p in planes]
    rec = [np.zeros_like(p) for p in org]
    ctus, ictus = np.zeros(n, CTU_DTYPE), np.zeros(n, CTU_INTER_DTYPE)
    po = (C.c_void_p * 3)(*[p.ctypes.data for p in org])
    pr = (C.c_void_p * 3)(*[p.ctypes.data for p in rec])
    L.hmo_compress_slice_inter.argtypes = [C.POINTER(Cfg), C.POINTER(InterSlice), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hmo_set_trace.argtypes = [C.c_char_p]
    if trace:
        L.hmo_set_trace(trace.encode())
    rc = L.hmo_compress_slice_inter(C.byref(cfg), C.byref(s), po, pr, ctus.ctypes.data, ictus.ctypes.data)
    if trace:
        L.hmo_set_trace(None)
    if rc != 0:
        raise RuntimeError(f"oracle (inter) failed rc={rc}")
    return rec, ctus, ictus


# ---- cu_qp_delta: adaptive QP / rate control ---------------------------------------------------------
class Dqp(C.Structure):
    _fields_ = [("use_dqp", C.c_int), ("dqp_flag_in", C.c_int), ("ctu_qp", C.c_void_p), ("qp_out", C.c_void_p), ("dqp_flag_out", C.POINTER(C.c_int)),
                ("ctu_lambda", C.c_void_p)]


def preanalyze(luma):
    """TEncPreanalyzer::xPreanalyze, layer 0: (activity per CTU in raster order, their average)"""
    L = lib()
    y = np.ascontiguousarray(luma, np.uint16)
    h, w = y.shape
    act = np.zeros(((w + 63) // 64) * ((h + 63) // 64), np.float64)
    avg = C.c_double(0)
    L.hmo_preanalyze.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
    L.hmo_preanalyze(y.ctypes.data, w, h, act.ctypes.data, C.byref(avg))
    return act, avg.value


def aq_qp(activity, avg_activity, aq_range, slice_qp, bit_depth):
    """TEncCu::xComputeQP for every unit -> int8 array"""
    L = lib()
    L.hmo_aq_qp.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    return np.array([L.hmo_aq_qp(float(a), float(avg_activity), int(aq_range), int(slice_qp), int(bit_depth)) for a in activity], np.int8)


def compress_dqp(planes, bit_depth, srec, finals, wpp, ctu_qp, dqp_flag_in, trace=None, ctu_lambda=None):
    """One slice (I, P or B: srec["slice_type"]) with cu_qp_delta enabled: ctu_qp = int8 QP per CTU (None: the slice QP everywhere, as the
    picture-level rate control runs it).  Returns (rec planes, ctus, inter ctus or None, qp (numCtus, 256) int8, dqp_flag_out)."""
    L = lib()
    h, w = planes[0].shape
    st = int(srec["slice_type"])
    cfg = Cfg(w, h, bit_depth, int(srec["qp"]), wpp, float(srec["lambda"]), float(srec["weight_cb"]))
    n = ((w + 63) // 64) * ((h + 63) // 64)
    s, keep = (None, None) if st == 2 else _inter_slice(srec, finals)
    org = [np.ascontiguousarray(p, np.uint16) for p in planes]
    rec = [np.zeros_like(p) for p in org]
    ctus, ictus = np.zeros(n, CTU_DTYPE), np.zeros(n, CTU_INTER_DTYPE)
    po = (C.c_void_p * 3)(*[p.ctypes.data for p in org])
    pr = (C.c_void_p * 3)(*[p.ctypes.data for p in rec])
    cq = np.ascontiguousarray(ctu_qp, np.int8) if ctu_qp is not None else None
    assert cq is None or len(cq) == n
    qp_out = np.zeros((n, 256), np.int8); flag_out = C.c_int(-1)
    cl = np.ascontiguousarray(ctu_lambda, np.float64) if ctu_lambda is not None else None      # the LCU-level rate control's lambda per CTU
    assert cl is None or len(cl) == n
    d = Dqp(1, int(dqp_flag_in), cq.ctypes.data if cq is not None else None, qp_out.ctypes.data, C.pointer(flag_out), cl.ctypes.data if cl is not None else None)
    L.hmo_compress_slice_dqp.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Dqp)]
    L.hmo_set_trace.argtypes = [C.c_char_p]
    if trace:
        L.hmo_set_trace(trace.encode())
    rc = L.hmo_compress_slice_dqp(C.byref(cfg), C.addressof(s) if s is not None else None, po, pr, ctus.ctypes.data, ictus.ctypes.data if st != 2 else None, C.byref(d))
    if trace:
        L.hmo_set_trace(None)
    if rc != 0:
        raise RuntimeError(f"oracle (dqp) failed rc={rc}")
    return rec, ctus, (ictus if st != 2 else None), qp_out, flag_out.value


def deblock(rec, bit_depth, qp, slice_type, ref_poc, ctus, ictus=None):
    """TComLoopFilter::loopFilterPic on the pre-deblocking reconstruction `rec` (3 planes); returns the filtered planes.
    ctus / ictus: arrays in CTU_DTYPE / CTU_INTER_DTYPE layout (ictus None for an I slice)."""
    L = lib()
    h, w = rec[0].shape
    cfg = Cfg(w, h, bit_depth, int(qp), 0, 1.0, 1.0)
    out = [np.ascontiguousarray(p, np.uint16).copy() for p in rec]
    pr = (C.c_void_p * 3)(*[p.ctypes.data for p in out])
    rp = np.ascontiguousarray(ref_poc, np.int32)
    c = np.ascontiguousarray(ctus)
    assert c.dtype == CTU_DTYPE
    ic = np.ascontiguousarray(ictus) if ictus is not None else None
    L.hmo_deblock.argtypes = [C.POINTER(Cfg), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.hmo_deblock(C.byref(cfg), int(slice_type), rp.ctypes.data, c.ctypes.data, ic.ctypes.data if ic is not None else None, pr)
    if rc != 0:
        raise RuntimeError(f"oracle deblock failed rc={rc}")
    return out


def sao(org, rec, bit_depth, qp, lam, chroma_weight, cabac_init_type, depth, disabled_rate):
    """TEncSampleAdaptiveOffset::SAOProcess on the deblocked planes `rec`; disabled_rate: float64 array (3, 8), updated in place.
    Returns (output planes, sao params int32 (numCtus, 3, 35), enabled flags)."""
    L = lib()
    h, w = rec[0].shape
    cfg = Cfg(w, h, bit_depth, int(qp), 0, float(lam), float(chroma_weight))
    o = [np.ascontiguousarray(p, np.uint16) for p in org]
    out = [np.ascontiguousarray(p, np.uint16).copy() for p in rec]
    po = (C.c_void_p * 3)(*[p.ctypes.data for p in o]); pr = (C.c_void_p * 3)(*[p.ctypes.data for p in out])
    n = ((w + 63) // 64) * ((h + 63) // 64)
    params = np.zeros((n, 3, 35), np.int32); en = np.zeros(3, np.int32)
    assert disabled_rate.dtype == np.float64 and disabled_rate.shape == (3, 8) and disabled_rate.flags["C_CONTIGUOUS"]
    L.hmo_sao.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.hmo_sao(C.byref(cfg), int(cabac_init_type), int(depth), disabled_rate.ctypes.data, po, pr, params.ctypes.data, en.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle sao failed rc={rc}")
    return out, params, en


class BitsSlice(C.Structure):
    _fields_ = [("slice_type", C.c_int), ("qp", C.c_int), ("cabac_init_type", C.c_int), ("num_ref_idx", C.c_int * 2),
                ("mvd_l1_zero", C.c_int), ("max_merge_cand", C.c_int), ("sao_enabled", C.c_int * 2)]


def encode_slice(width, height, bit_depth, wpp, slice_type, qp, ctus, ictus=None, cabac_init_type=None, num_ref_idx=(0, 0), mvd_l1_zero=0,
                 max_merge_cand=5, sao=None, sao_enabled=(0, 0)):
    """TEncSlice::encodeSlice: the CABAC-coded slice data of one picture.  ctus / ictus as compress*() return them; sao = int32 (numCtus, 3, 35)
    as sao() returns it.  Returns (list of substream byte strings, next cabac_init_type, number of bins)."""
    L = lib()
    cfg = Cfg(width, height, bit_depth, int(qp), int(wpp), 0.0, 0.0)
    s = BitsSlice()
    s.slice_type, s.qp = int(slice_type), int(qp)
    s.cabac_init_type = int(slice_type if cabac_init_type is None else cabac_init_type)
    s.num_ref_idx[0], s.num_ref_idx[1] = int(num_ref_idx[0]), int(num_ref_idx[1])
    s.mvd_l1_zero, s.max_merge_cand = int(mvd_l1_zero), int(max_merge_cand)
    s.sao_enabled[0], s.sao_enabled[1] = int(sao_enabled[0]), int(sao_enabled[1])
    c = np.ascontiguousarray(ctus)
    assert c.dtype == CTU_DTYPE
    ic = np.ascontiguousarray(ictus) if ictus is not None else None
    sp = np.ascontiguousarray(sao, np.int32) if sao is not None else None
    n_sub = (height + 63) // 64 if wpp else 1
    cap = 4 * width * height + 4096
    out = np.zeros(cap, np.uint8); sizes = np.zeros(n_sub, np.uint32)
    nxt = C.c_int(-1); bins = C.c_uint32(0)
    L.hmo_encode_slice.argtypes = [C.POINTER(Cfg), C.POINTER(BitsSlice), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    rc = L.hmo_encode_slice(C.byref(cfg), C.byref(s), c.ctypes.data, ic.ctypes.data if ic is not None else None,
                            sp.ctypes.data if sp is not None else None, out.ctypes.data, cap, sizes.ctypes.data, C.byref(nxt), C.byref(bins))
    if rc != 0:
        raise RuntimeError(f"oracle encode_slice failed rc={rc}")
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    raw = out.tobytes()
    return [raw[offs[k]:offs[k + 1]] for k in range(n_sub)], nxt.value, bins.value


def yuv_read(raw, file_w, file_h, file_bd, internal_bd, pad_x=0, pad_y=0):
    """TVideoIOYuv::read of one frame: raw bytes -> 3 planes (uint16) of (file_w + pad_x) x (file_h + pad_y)"""
    L = lib()
    w, h = file_w + pad_x, file_h + pad_y
    planes = [np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16)]
    buf = np.frombuffer(raw, np.uint8)
    pp = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    L.hmo_yuv_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.hmo_yuv_read(buf.ctypes.data, file_w, file_h, file_bd, internal_bd, pad_x, pad_y, pp)
    if rc != 0:
        raise RuntimeError(f"oracle yuv_read failed rc={rc}")
    return planes


def yuv_write(planes, internal_bd, file_bd, crop_right=0, crop_bottom=0):
    """TVideoIOYuv::write of one frame: 3 planes (uint16) -> raw bytes of the cropped picture at file_bd"""
    L = lib()
    h, w = planes[0].shape
    pl = [np.ascontiguousarray(p, np.uint16) for p in planes]
    n = (w - crop_right) * (h - crop_bottom) * 3 // 2 * (2 if file_bd > 8 else 1)
    out = np.zeros(n, np.uint8)
    pp = (C.c_void_p * 3)(*[p.ctypes.data for p in pl])
    L.hmo_yuv_write.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.hmo_yuv_write(pp, w, h, internal_bd, file_bd, crop_right, crop_bottom, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle yuv_write failed rc={rc}")
    return out.tobytes()
