"""ctypes binding of libhm355.so (the C ABI in include/hm355.h) plus small test/bench helpers.

PyTorch is not needed here: the library owns its device buffers.  The binding fails loudly when the
HIP library is missing or no GPU is usable -- there is no CPU fallback in the product path.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhm355.so")


class SeqCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "bit_depth", "ctu_size", "max_cu_depth", "tu_log2_max",
                                          "tu_log2_min", "tu_max_depth_intra", "wavefront_synchro", "max_batch")]


class SliceDesc(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("qp", C.c_int32), ("lambda_", C.c_double), ("chroma_weight", C.c_double)]


class Planes(C.Structure):
    _fields_ = [("plane", C.POINTER(C.c_uint16) * 3)]


class CtuOut(C.Structure):
    _fields_ = [("total_cost", C.c_double), ("total_bits", C.c_uint32), ("total_dist", C.c_uint32),
                ("depth", C.c_uint8 * 256), ("part_size", C.c_uint8 * 256), ("pred_mode", C.c_uint8 * 256),
                ("intra_dir_luma", C.c_uint8 * 256), ("intra_dir_chroma", C.c_uint8 * 256), ("tr_idx", C.c_uint8 * 256),
                ("cbf", (C.c_uint8 * 256) * 3), ("tskip", (C.c_uint8 * 256) * 3),
                ("coeff_y", C.c_int32 * 4096), ("coeff_cb", C.c_int32 * 1024), ("coeff_cr", C.c_int32 * 1024)]


class SliceStats(C.Structure):
    _fields_ = [("pic_total_bits", C.c_uint64), ("pic_rd_cost", C.c_double), ("pic_dist", C.c_uint64)]


CTU_DTYPE = np.dtype([("total_cost", "<f8"), ("total_bits", "<u4"), ("total_dist", "<u4"),
                      ("depth", "u1", 256), ("part_size", "u1", 256), ("pred_mode", "u1", 256),
                      ("intra_dir_luma", "u1", 256), ("intra_dir_chroma", "u1", 256), ("tr_idx", "u1", 256),
                      ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
                      ("coeff_y", "<i4", 4096), ("coeff_cb", "<i4", 1024), ("coeff_cr", "<i4", 1024)])
assert CTU_DTYPE.itemsize == C.sizeof(CtuOut)

class RefPic(C.Structure):
    _fields_ = [("poc", C.c_int32), ("slice_type", C.c_int32), ("long_term", C.c_int32), ("plane", C.c_void_p * 3),
                ("pred_mode", C.c_void_p), ("mv", C.c_void_p * 2), ("ref_idx", C.c_void_p * 2),
                ("num_ref", C.c_int32 * 2), ("ref_poc", (C.c_int32 * 16) * 2), ("ref_lt", (C.c_int32 * 16) * 2)]


class InterSliceDesc(C.Structure):
    _fields_ = [("base", SliceDesc), ("poc", C.c_int32), ("cabac_init_type", C.c_int32), ("num_ref_idx", C.c_int32 * 2),
                ("ref", (C.POINTER(RefPic) * 16) * 2),
                ("col_from_l0", C.c_int32), ("col_ref_idx", C.c_int32), ("tmvp", C.c_int32), ("mvd_l1_zero", C.c_int32),
                ("max_merge_cand", C.c_int32), ("check_ldc", C.c_int32),
                ("lambda_motion_sad", C.c_uint32), ("lambda_motion_sse", C.c_uint32), ("dev_ref", (C.c_void_p * 16) * 2)]


CTU_INTER_DTYPE = np.dtype([("skip", "u1", 256), ("merge_flag", "u1", 256), ("merge_idx", "u1", 256), ("inter_dir", "u1", 256),
                            ("mv", "<i2", (2, 256, 2)), ("mvd", "<i2", (2, 256, 2)),
                            ("ref_idx", "i1", (2, 256)), ("mvp_idx", "i1", (2, 256)), ("mvp_num", "i1", (2, 256))])

class DbkDesc(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("qp", C.c_int32), ("ref_poc", (C.c_int32 * 16) * 2)]


class SaoDesc(C.Structure):
    _fields_ = [("qp", C.c_int32), ("cabac_init_type", C.c_int32), ("depth", C.c_int32), ("lambda_", C.c_double), ("chroma_weight", C.c_double),
                ("disabled_rate", (C.c_double * 8) * 3), ("enabled", C.c_int32 * 3), ("params", C.c_void_p)]


class DqpDesc(C.Structure):
    _fields_ = [("use_dqp", C.c_int32), ("dqp_flag_in", C.c_int32), ("ctu_qp", C.c_void_p)]


class BitsDesc(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("qp", C.c_int32), ("cabac_init_type", C.c_int32), ("num_ref_idx", C.c_int32 * 2), ("mvd_l1_zero", C.c_int32),
                ("max_merge_cand", C.c_int32), ("sao_enabled", C.c_int32 * 2), ("out", C.c_void_p), ("out_cap", C.c_size_t), ("sub_sizes", C.c_void_p),
                ("next_cabac_init_type", C.c_int32), ("num_bins", C.c_uint32)]


EXPORTS = ["hm355_build_id", "hm355_set_dqp", "hm355_get_dqp", "hm355_preanalyze", "hm355_create", "hm355_destroy", "hm355_last_error", "hm355_compress_slice", "hm355_compress_slices",
           "hm355_compress_slice_inter", "hm355_compress_slices_inter", "hm355_deblock", "hm355_deblock_run", "hm355_ref_from_slot", "hm355_ref_release", "hm355_ref_bytes", "hm355_ref_export", "hm355_ref_import", "hm355_sao_run",
           "hm355_num_substreams", "hm355_encode_slices_run", "hm355_encode_slice",
           "hm355_upload_file_frames", "hm355_download_file_frames", "hm355_download_org",
           "hm355_upload", "hm355_run", "hm355_run_begin", "hm355_run_wait", "hm355_set_lane_share", "hm355_run_rows", "hm355_boundary_bytes", "hm355_export_boundary", "hm355_import_boundary", "hm355_download", "hm355_last_run_info", "hm355_dist_batch",
           "hm355_transform_batch"]


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hm355 has no CPU fallback)")
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            if os.environ.get("HM355_OLD_LIB_OK"):      # A/B timing runs against a library built from an earlier commit (tools/quick_timing.py)
                setattr(lib, name, lib.hm355_last_run_info)
                continue
            raise AttributeError(f"{path}: {name} is not exported")
    lib.hm355_build_id.restype = C.c_char_p
    lib.hm355_create.argtypes = [C.POINTER(SeqCfg), C.POINTER(C.c_void_p)]
    lib.hm355_destroy.argtypes = [C.c_void_p]
    lib.hm355_last_error.argtypes = [C.c_void_p]
    lib.hm355_last_error.restype = C.c_char_p
    lib.hm355_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Planes)]
    lib.hm355_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(SliceDesc)]
    lib.hm355_run_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(SliceDesc)]
    lib.hm355_run_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    lib.hm355_set_lane_share.argtypes = [C.c_void_p, C.c_int]
    lib.hm355_set_dqp.argtypes = [C.c_void_p, C.c_int, C.POINTER(DqpDesc)]
    lib.hm355_get_dqp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int32)]
    lib.hm355_preanalyze.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.hm355_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(Planes), C.c_void_p, C.POINTER(SliceStats)]
    lib.hm355_run_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(SliceDesc), C.c_int, C.c_int]
    lib.hm355_boundary_bytes.argtypes = [C.c_void_p]
    lib.hm355_boundary_bytes.restype = C.c_size_t
    lib.hm355_export_boundary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.hm355_import_boundary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.hm355_last_run_info.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.hm355_compress_slice.argtypes = [C.c_void_p, C.POINTER(SliceDesc), C.POINTER(Planes), C.POINTER(Planes), C.c_void_p,
                                         C.POINTER(SliceStats)]
    lib.hm355_compress_slice_inter.argtypes = [C.c_void_p, C.POINTER(InterSliceDesc), C.POINTER(Planes), C.POINTER(Planes), C.c_void_p,
                                               C.c_void_p, C.POINTER(SliceStats)]
    lib.hm355_compress_slices_inter.argtypes = [C.c_void_p, C.c_int, C.POINTER(InterSliceDesc), C.POINTER(Planes), C.POINTER(Planes),
                                                C.c_void_p, C.c_void_p, C.POINTER(SliceStats)]
    lib.hm355_deblock.argtypes = [C.c_void_p, C.POINTER(DbkDesc), C.c_void_p, C.c_void_p, C.POINTER(Planes)]
    lib.hm355_deblock_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(DbkDesc)]
    lib.hm355_ref_from_slot.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hm355_ref_release.argtypes = [C.c_void_p, C.c_void_p]
    lib.hm355_ref_release.restype = None
    lib.hm355_ref_bytes.argtypes = [C.c_void_p]
    lib.hm355_ref_bytes.restype = C.c_size_t
    lib.hm355_ref_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hm355_ref_import.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    lib.hm355_sao_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(SaoDesc)]
    lib.hm355_upload_file_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
    lib.hm355_download_file_frames.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.hm355_download_org.argtypes = [C.c_void_p, C.c_int, C.POINTER(Planes)]
    lib.hm355_num_substreams.argtypes = [C.c_void_p]
    lib.hm355_encode_slices_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(BitsDesc)]
    lib.hm355_encode_slice.argtypes = [C.c_void_p, C.POINTER(BitsDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hm355_dist_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hm355_transform_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def intra_lambda(qp):
    """I-slice lambda / chroma distortion weight of an all-intra GOP
    (TEncSlice::initEncSlice, TEncSlice.cpp:323-352; setUpLambda :132-159)."""
    chroma_scale = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                    29, 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48,
                    49, 50, 51]
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    qpc = chroma_scale[min(max(qp, 0), 57)]
    return lam, 2.0 ** ((qp - qpc) / 3.0)


def aq_activities(sums):
    """TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:117-137) from the per-CTU quadrant sums hm355_preanalyze returns: activity per CTU
    (1 + the smallest quadrant "variance", every quadrant's sums divided by the sample count of the WHOLE unit, as the reference does) and
    the average (running sum in raster order / number of units)."""
    import math
    act = np.zeros(len(sums), np.float64)
    total = 0.0
    for a, s in enumerate(sums):
        n = int(s[8])
        mv = math.inf
        for k in range(4):
            avg = float(int(s[k])) / n
            mv = min(mv, float(int(s[4 + k])) / n - avg * avg)
        act[a] = 1.0 + mv
        total += act[a]
    return act, total / len(sums)


def aq_ctu_qp(act, avg, aq_range, slice_qp, bit_depth):
    """TEncCu::xComputeQP (TEncCu.cpp:1154-1176) for every CTU -> int8 array"""
    import math
    out = np.zeros(len(act), np.int8)
    max_q_scale = math.pow(2.0, aq_range / 6.0)
    for a, d in enumerate(act):
        norm = (max_q_scale * d + avg) / (d + max_q_scale * avg)
        off = int(math.floor(math.log(norm) / math.log(2.0) * 6.0 + 0.49999))
        out[a] = min(51, max(-6 * (bit_depth - 8), slice_qp + off))
    return out


def _planes(arrs):
    p = Planes()
    for k in range(3):
        a = arrs[k]
        assert a.dtype == np.uint16 and a.flags["C_CONTIGUOUS"]
        p.plane[k] = a.ctypes.data_as(C.POINTER(C.c_uint16))
    return p


class Encoder:
    """Thin object wrapper of a hm355_ctx (one per TEncTop-like encoder instance)."""

    def __init__(self, width, height, bit_depth, wpp=0, max_batch=1, lib=None):
        self.lib = lib or load_library()
        self.w, self.h, self.bd, self.wpp, self.max_batch = width, height, bit_depth, wpp, max_batch
        self.num_ctus = ((width + 63) // 64) * ((height + 63) // 64)
        cfg = SeqCfg(width, height, bit_depth, 64, 4, 5, 2, 3, wpp, max_batch)
        h = C.c_void_p()
        rc = self.lib.hm355_create(C.byref(cfg), C.byref(h))
        self.h_ = h
        if rc != 0:
            msg = self.lib.hm355_last_error(h).decode() if h else ""
            if h:
                self.lib.hm355_destroy(h)
                self.h_ = None
            raise RuntimeError(f"hm355_create failed rc={rc} {msg}")

    def close(self):
        if getattr(self, "h_", None):
            self.lib.hm355_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed rc={rc}: {self.lib.hm355_last_error(self.h_).decode()}")

    def upload(self, slot, planes):
        p = _planes(planes)
        self._check(self.lib.hm355_upload(self.h_, slot, C.byref(p)), "hm355_upload")

    def run(self, n, qp):
        lam, cw = intra_lambda(qp)
        sl = (SliceDesc * n)(*[SliceDesc(2, qp, lam, cw) for _ in range(n)])
        self._check(self.lib.hm355_run(self.h_, n, sl), "hm355_run")
        ms, launches = C.c_double(), C.c_int()
        self.lib.hm355_last_run_info(self.h_, C.byref(ms), C.byref(launches))
        return ms.value, launches.value

    def set_dqp(self, slot, ctu_qp=None, dqp_flag_in=0, use_dqp=1):
        """hm355_set_dqp: the following searches / deblocking / bitstream pass of the slot run with cu_qp_delta; ctu_qp int8 [numCtus] or None"""
        q = np.ascontiguousarray(ctu_qp, np.int8) if ctu_qp is not None else None
        assert q is None or len(q) == self.num_ctus
        d = DqpDesc(int(use_dqp), int(dqp_flag_in), q.ctypes.data if q is not None else None)
        self._check(self.lib.hm355_set_dqp(self.h_, slot, C.byref(d)), "hm355_set_dqp")

    def get_dqp(self, slot):
        """hm355_get_dqp -> (m_phQP int8 (numCtus, 256), m_bEncodeDQP after the slice)"""
        qp = np.zeros((self.num_ctus, 256), np.int8); f = C.c_int32(-1)
        self._check(self.lib.hm355_get_dqp(self.h_, slot, qp.ctypes.data, C.byref(f)), "hm355_get_dqp")
        return qp, f.value

    def preanalyze(self, slot):
        """hm355_preanalyze -> uint64 (numCtus, 9): the eight quadrant sums of every CTU and, last, the number of samples of the unit inside the picture"""
        s = np.zeros((self.num_ctus, 8), np.uint64)
        self._check(self.lib.hm355_preanalyze(self.h_, slot, s.ctypes.data), "hm355_preanalyze")
        wc = (self.w + 63) // 64
        n = np.array([min(64, self.w - (a % wc) * 64) * min(64, self.h - (a // wc) * 64) for a in range(self.num_ctus)], np.uint64)
        return np.concatenate([s, n[:, None]], axis=1)

    def run_begin(self, lane, first_slot, n, qp):
        """enqueue the search over slots [first_slot, first_slot + n) on pipeline lane `lane`; returns at once (hm355_run_begin)"""
        lam, cw = intra_lambda(qp)
        sl = (SliceDesc * n)(*[SliceDesc(2, qp, lam, cw) for _ in range(n)])
        self._check(self.lib.hm355_run_begin(self.h_, lane, first_slot, n, sl), "hm355_run_begin")

    def set_lane_share(self, launches_in_flight):
        self._check(self.lib.hm355_set_lane_share(self.h_, int(launches_in_flight)), "hm355_set_lane_share")

    def run_wait(self, lane):
        """wait for the launch of `lane`; returns its kernel time in ms (hm355_run_wait)"""
        ms = C.c_double()
        self._check(self.lib.hm355_run_wait(self.h_, lane, C.byref(ms)), "hm355_run_wait")
        return ms.value

    def run_rows(self, first_slot, n, qp, first_row, last_row):
        """the search over CTU rows [first_row, last_row] of the pictures in slots [first_slot, first_slot + n) (a band; see hm355_run_rows)"""
        lam, cw = intra_lambda(qp)
        sl = (SliceDesc * n)(*[SliceDesc(2, qp, lam, cw) for _ in range(n)])
        self._check(self.lib.hm355_run_rows(self.h_, first_slot, n, sl, first_row, last_row), "hm355_run_rows")
        ms, launches = C.c_double(), C.c_int()
        self.lib.hm355_last_run_info(self.h_, C.byref(ms), C.byref(launches))
        return ms.value, launches.value

    def boundary_bytes(self):
        return int(self.lib.hm355_boundary_bytes(self.h_))

    def export_boundary(self, slot, row):
        """what the band below needs from CTU row `row` of the picture in `slot`, as a uint8 array"""
        buf = np.empty(self.boundary_bytes(), np.uint8)
        self._check(self.lib.hm355_export_boundary(self.h_, slot, row, buf.ctypes.data), "hm355_export_boundary")
        return buf

    def import_boundary(self, slot, row, data):
        buf = np.ascontiguousarray(data, np.uint8)
        assert buf.size == self.boundary_bytes()
        self._check(self.lib.hm355_import_boundary(self.h_, slot, row, buf.ctypes.data), "hm355_import_boundary")

    def export_boundary_ptr(self, slot, row, ptr):
        """the same into boundary_bytes() bytes at address `ptr` -- host or device memory"""
        self._check(self.lib.hm355_export_boundary(self.h_, slot, row, C.c_void_p(ptr)), "hm355_export_boundary")

    def import_boundary_ptr(self, slot, row, ptr):
        self._check(self.lib.hm355_import_boundary(self.h_, slot, row, C.c_void_p(ptr)), "hm355_import_boundary")

    def download(self, slot, want_ctus=True):
        rec = [np.zeros((self.h, self.w), np.uint16), np.zeros((self.h // 2, self.w // 2), np.uint16),
               np.zeros((self.h // 2, self.w // 2), np.uint16)]
        p = _planes(rec)
        ctus = np.zeros(self.num_ctus, CTU_DTYPE) if want_ctus else None
        st = SliceStats()
        self._check(self.lib.hm355_download(self.h_, slot, C.byref(p), ctus.ctypes.data if want_ctus else None, C.byref(st)),
                    "hm355_download")
        return rec, ctus, (st.pic_total_bits, st.pic_rd_cost, st.pic_dist)

    def compress(self, frames, qp):
        """frames: list of (Y,U,V) uint16 planes -> list of (rec, ctus, stats)"""
        for i, f in enumerate(frames):
            self.upload(i, f)
        self.run(len(frames), qp)
        return [self.download(i) for i in range(len(frames))]

    def compress_inter_batch(self, jobs):
        """n independent P slices through hm355_compress_slices_inter.  jobs: list of (planes, slice_params, ref_pics) with
        slice_params: dict with qp, lambda, chroma_weight, poc, cabac_init_type, num_ref_idx, ref_poc (per list), col_from_l0,
        col_ref_idx, tmvp, mvd_l1_zero, max_merge_cand, check_ldc, lambda_motion_sad, lambda_motion_sse;
        ref_pics: {poc: dict(slice_type, rec=[3 planes], pred_mode, mv=[2], ref_idx=[2], num_ref_idx, ref_poc, ref_long_term)}
        (a ref_pics dict object shared by several jobs is uploaded once).  Returns [(rec planes, ctus, inter ctus, stats)]."""
        n = len(jobs)
        keep, conv = [], {}
        descs = (InterSliceDesc * n)()
        orgs, recs, po, pr = [], [], (Planes * n)(), (Planes * n)()
        ctus = [np.zeros(self.num_ctus, CTU_DTYPE) for _ in range(n)]
        ictus = [np.zeros(self.num_ctus, CTU_INTER_DTYPE) for _ in range(n)]
        for k, (planes, sp, ref_pics) in enumerate(jobs):
            if id(ref_pics) not in conv:
                refs = {}
                for poc, f in ref_pics.items():
                    if "dev" in f:                           # device-resident reference (ref_from_slot)
                        refs[int(poc)] = f["dev"]
                        continue
                    pl = [np.ascontiguousarray(p, np.uint16) for p in f["rec"]]
                    pm = np.ascontiguousarray(f["pred_mode"], np.uint8)
                    mv = [np.ascontiguousarray(f["mv"][l], np.int16) for l in range(2)]
                    ri = [np.ascontiguousarray(f["ref_idx"][l], np.int8) for l in range(2)]
                    keep += pl + [pm] + mv + ri
                    r = RefPic()
                    r.poc, r.slice_type, r.long_term = int(poc), int(f["slice_type"]), 0
                    for c in range(3):
                        r.plane[c] = pl[c].ctypes.data
                    r.pred_mode = pm.ctypes.data
                    for l in range(2):
                        r.mv[l] = mv[l].ctypes.data; r.ref_idx[l] = ri[l].ctypes.data; r.num_ref[l] = int(f["num_ref_idx"][l])
                        for i in range(16):
                            r.ref_poc[l][i] = int(f["ref_poc"][l][i]); r.ref_lt[l][i] = int(f["ref_long_term"][l][i])
                    refs[int(poc)] = r
                conv[id(ref_pics)] = refs
            refs = conv[id(ref_pics)]
            s = descs[k]
            s.base = SliceDesc(int(sp.get("slice_type", 1)), int(sp["qp"]), float(sp["lambda"]), float(sp["chroma_weight"]))
            s.poc, s.cabac_init_type = int(sp["poc"]), int(sp["cabac_init_type"])
            for l in range(2):
                s.num_ref_idx[l] = int(sp["num_ref_idx"][l])
                for i in range(s.num_ref_idx[l]):
                    rr = refs[int(sp["ref_poc"][l][i])]
                    if isinstance(rr, RefPic):
                        s.ref[l][i] = C.pointer(rr)
                    else:
                        s.dev_ref[l][i] = rr
            for key in ("col_from_l0", "col_ref_idx", "tmvp", "mvd_l1_zero", "max_merge_cand", "check_ldc", "lambda_motion_sad",
                        "lambda_motion_sse"):
                setattr(s, key, int(sp[key]))
            org = [np.ascontiguousarray(p, np.uint16) for p in planes]
            rec = [np.zeros_like(p) for p in org]
            orgs.append(org); recs.append(rec)
            po[k], pr[k] = _planes(org), _planes(rec)
        pc = (C.c_void_p * n)(*[a.ctypes.data for a in ctus])
        pi = (C.c_void_p * n)(*[a.ctypes.data for a in ictus])
        st = (SliceStats * n)()
        self._check(self.lib.hm355_compress_slices_inter(self.h_, n, descs, po, pr, pc, pi, st), "hm355_compress_slices_inter")
        del keep
        return [(recs[k], ctus[k], ictus[k], (st[k].pic_total_bits, st[k].pic_rd_cost, st[k].pic_dist)) for k in range(n)]

    def compress_inter(self, planes, slice_params, ref_pics):
        """One P slice (see compress_inter_batch)."""
        return self.compress_inter_batch([(planes, slice_params, ref_pics)])[0]

    @staticmethod
    def _dbk_desc(slice_type, qp, ref_poc):
        d = DbkDesc(int(slice_type), int(qp))
        for l in range(2):
            for i in range(16):
                d.ref_poc[l][i] = int(ref_poc[l][i]) if ref_poc is not None else 0
        return d

    def deblock(self, rec, slice_type, qp, ref_poc, ctus, ictus=None):
        """hm355_deblock: TComLoopFilter::loopFilterPic on host buffers; returns the filtered planes"""
        out = [np.ascontiguousarray(p, np.uint16).copy() for p in rec]
        pr = _planes(out)
        d = self._dbk_desc(slice_type, qp, ref_poc)
        c = np.ascontiguousarray(ctus)
        assert c.dtype == CTU_DTYPE
        ic = np.ascontiguousarray(ictus) if ictus is not None else None
        self._check(self.lib.hm355_deblock(self.h_, C.byref(d), c.ctypes.data, ic.ctypes.data if ic is not None else None, C.byref(pr)), "hm355_deblock")
        return out

    def ref_from_slot(self, slot, poc, is_inter, num_ref_idx=(0, 0), ref_poc=None, ref_long_term=None):
        """hm355_ref_from_slot: the slot's (deblocked) picture as a device-resident reference; returns {"dev": handle} for compress_inter"""
        nr = (C.c_int32 * 2)(int(num_ref_idx[0]), int(num_ref_idx[1]))
        rp = np.ascontiguousarray(ref_poc if ref_poc is not None else np.zeros((2, 16)), np.int32)
        rl = np.ascontiguousarray(ref_long_term if ref_long_term is not None else np.zeros((2, 16)), np.int32)
        h = C.c_void_p()
        self._check(self.lib.hm355_ref_from_slot(self.h_, slot, int(poc), int(bool(is_inter)), nr, rp.ctypes.data, rl.ctypes.data, C.byref(h)),
                    "hm355_ref_from_slot")
        return {"dev": h.value}

    def ref_bytes(self):
        return int(self.lib.hm355_ref_bytes(self.h_))

    def ref_export(self, ref, ptr=None, user=(0.0, 0.0, 0.0, 0.0)):
        """hm355_ref_export: the reference picture as one blob; into ref_bytes() bytes at `ptr` (host or device memory), or into a new uint8 array"""
        u = (C.c_double * 4)(*[float(v) for v in user])
        buf = None
        if ptr is None:
            buf = np.empty(self.ref_bytes(), np.uint8); ptr = buf.ctypes.data
        self._check(self.lib.hm355_ref_export(self.h_, ref["dev"], C.c_void_p(ptr), u), "hm355_ref_export")
        return buf

    def ref_import(self, data):
        """hm355_ref_import: blob (uint8 array, or an address of host / device memory) -> ({"dev": handle}, the four user doubles)"""
        u = (C.c_double * 4)()
        if not isinstance(data, int):
            data = np.ascontiguousarray(data, np.uint8); assert data.size == self.ref_bytes()
        h = C.c_void_p()
        self._check(self.lib.hm355_ref_import(self.h_, C.c_void_p(data if isinstance(data, int) else data.ctypes.data), C.byref(h), u), "hm355_ref_import")
        return {"dev": h.value}, tuple(u)

    def ref_release(self, ref):
        self.lib.hm355_ref_release(self.h_, ref["dev"])

    def sao_run(self, descs):
        """hm355_sao_run on slots 0..n-1 (deblocked pictures + originals resident).  descs: list of dicts with qp, lambda, chroma_weight,
        cabac_init_type, depth, disabled_rate (float64 array (3, 8), updated in place).  Returns [(enabled flags, params (numCtus, 3, 35))]."""
        n = len(descs)
        arr = (SaoDesc * n)()
        params = [np.zeros((self.num_ctus, 3, 35), np.int32) for _ in range(n)]
        for k, d in enumerate(descs):
            a = arr[k]
            a.qp, a.cabac_init_type, a.depth = int(d["qp"]), int(d["cabac_init_type"]), int(d["depth"])
            a.lambda_, a.chroma_weight = float(d["lambda"]), float(d["chroma_weight"])
            for c in range(3):
                for t in range(8):
                    a.disabled_rate[c][t] = float(d["disabled_rate"][c][t])
            a.params = params[k].ctypes.data
        self._check(self.lib.hm355_sao_run(self.h_, n, arr), "hm355_sao_run")
        out = []
        for k, d in enumerate(descs):
            for c in range(3):
                for t in range(8):
                    d["disabled_rate"][c][t] = arr[k].disabled_rate[c][t]
            out.append((tuple(int(v) for v in arr[k].enabled), params[k]))
        return out

    def upload_file_frames(self, frames, file_w, file_h, file_bd):
        """hm355_upload_file_frames: raw 4:2:0 frames (bytes, as on disk) -> original planes of slots 0..n-1 (TVideoIOYuv::read on the device).
        Returns the kernel time in ms."""
        n = len(frames)
        bufs = [np.frombuffer(f, np.uint8) for f in frames]
        need = file_w * file_h * 3 // 2 * (2 if file_bd > 8 else 1)
        assert all(len(b) == need for b in bufs)
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        self._check(self.lib.hm355_upload_file_frames(self.h_, n, ptrs, file_w, file_h, file_bd), "hm355_upload_file_frames")
        ms, launches = C.c_double(), C.c_int()
        self.lib.hm355_last_run_info(self.h_, C.byref(ms), C.byref(launches))
        return ms.value

    def download_file_frames(self, n, file_bd, conf_right=0, conf_bottom=0, source=0):
        """hm355_download_file_frames: the reconstruction (source 0) or the original planes (source 1) of slots 0..n-1 as raw 4:2:0 frames
        (TVideoIOYuv::write on the device).  Returns (list of bytes, kernel ms)."""
        size = (self.w - conf_right) * (self.h - conf_bottom) * 3 // 2 * (2 if file_bd > 8 else 1)
        bufs = [np.zeros(size, np.uint8) for _ in range(n)]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        self._check(self.lib.hm355_download_file_frames(self.h_, n, ptrs, file_bd, conf_right, conf_bottom, source), "hm355_download_file_frames")
        ms, launches = C.c_double(), C.c_int()
        self.lib.hm355_last_run_info(self.h_, C.byref(ms), C.byref(launches))
        return [b.tobytes() for b in bufs], ms.value

    def download_org(self, slot):
        """the original planes of a slot as the encoder sees them"""
        planes = [np.zeros((self.h, self.w), np.uint16), np.zeros((self.h // 2, self.w // 2), np.uint16), np.zeros((self.h // 2, self.w // 2), np.uint16)]
        p = _planes(planes)
        self._check(self.lib.hm355_download_org(self.h_, slot, C.byref(p)), "hm355_download_org")
        return planes

    def _bits_descs(self, descs):
        n = len(descs)
        nsub = self.lib.hm355_num_substreams(self.h_)
        arr = (BitsDesc * n)()
        cap = 4 * self.w * self.h + 4096
        outs = [np.zeros(cap, np.uint8) for _ in range(n)]; sizes = [np.zeros(nsub, np.uint32) for _ in range(n)]
        for k, d in enumerate(descs):
            a = arr[k]
            a.slice_type, a.qp = int(d["slice_type"]), int(d["qp"])
            a.cabac_init_type = int(d.get("cabac_init_type", d["slice_type"]) if d.get("cabac_init_type") is not None else d["slice_type"])
            nri = d.get("num_ref_idx", (0, 0))
            a.num_ref_idx[0], a.num_ref_idx[1] = int(nri[0]), int(nri[1])
            a.mvd_l1_zero, a.max_merge_cand = int(d.get("mvd_l1_zero", 0)), int(d.get("max_merge_cand", 5))
            en = d.get("sao_enabled", (0, 0))
            a.sao_enabled[0], a.sao_enabled[1] = int(en[0]), int(en[1])
            a.out, a.out_cap, a.sub_sizes = outs[k].ctypes.data, cap, sizes[k].ctypes.data
        return arr, outs, sizes

    @staticmethod
    def _bits_results(arr, outs, sizes):
        res = []
        for k in range(len(outs)):
            offs = np.concatenate([[0], np.cumsum(sizes[k])]).astype(int)
            raw = outs[k][:offs[-1]].tobytes()
            res.append(([raw[offs[i]:offs[i + 1]] for i in range(len(sizes[k]))], int(arr[k].next_cabac_init_type), int(arr[k].num_bins)))
        return res

    def encode_slices_run(self, descs):
        """hm355_encode_slices_run on slots 0..n-1 (search results, and SAO parameters when sao_enabled is set, resident).  descs: dicts with
        slice_type, qp and, for P / B slices, cabac_init_type, num_ref_idx, mvd_l1_zero, max_merge_cand; sao_enabled (luma, chroma).
        Returns [(substream byte strings, next cabac_init_type, number of bins)]."""
        arr, outs, sizes = self._bits_descs(descs)
        self._check(self.lib.hm355_encode_slices_run(self.h_, len(descs), arr), "hm355_encode_slices_run")
        return self._bits_results(arr, outs, sizes)

    def encode_slice(self, desc, ctus, ictus=None, sao=None):
        """hm355_encode_slice: host buffers in (CTU data of one slice, optional motion data and SAO parameters (numCtus, 3, 35) int32)"""
        arr, outs, sizes = self._bits_descs([desc])
        c = np.ascontiguousarray(ctus); assert c.dtype == CTU_DTYPE and len(c) == self.num_ctus
        ic = np.ascontiguousarray(ictus) if ictus is not None else None
        sp = np.ascontiguousarray(sao, np.int32) if sao is not None else None
        self._check(self.lib.hm355_encode_slice(self.h_, arr, c.ctypes.data, ic.ctypes.data if ic is not None else None,
                                                sp.ctypes.data if sp is not None else None), "hm355_encode_slice")
        return self._bits_results(arr, outs, sizes)[0]

    def deblock_run(self, descs):
        """hm355_deblock_run on slots 0..n-1 (device-resident); descs: list of (slice_type, qp, ref_poc).  Returns kernel ms."""
        n = len(descs)
        arr = (DbkDesc * n)(*[self._dbk_desc(*d) for d in descs])
        self._check(self.lib.hm355_deblock_run(self.h_, n, arr), "hm355_deblock_run")
        ms, launches = C.c_double(), C.c_int()
        self.lib.hm355_last_run_info(self.h_, C.byref(ms), C.byref(launches))
        return ms.value

    def dist_batch(self, kind, org, cur, bit_depth):
        """org/cur: (count, n, n) int16; kind 0 SAD, 1 SSE, 2 SATD, 3 SAD with row sub-sampling"""
        count, n = org.shape[0], org.shape[1]
        out = np.zeros(count, np.uint32)
        o, c = np.ascontiguousarray(org, np.int16), np.ascontiguousarray(cur, np.int16)
        self._check(self.lib.hm355_dist_batch(self.h_, kind, n, bit_depth, count, o.ctypes.data, c.ctypes.data, out.ctypes.data),
                    "hm355_dist_batch")
        return out

    def transform_batch(self, inverse, blocks, bit_depth, use_dst=0):
        count, n = blocks.shape[0], blocks.shape[1]
        b = np.ascontiguousarray(blocks, np.int32)
        out = np.zeros_like(b)
        self._check(self.lib.hm355_transform_batch(self.h_, inverse, n, bit_depth, use_dst, count, b.ctypes.data, out.ctypes.data),
                    "hm355_transform_batch")
        return out
