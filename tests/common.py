import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META_FIELDS = ["depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx", "cbf", "tskip"]
CASES = ["c1_416x240_8b_qp32", "wpp_416x240_10b_qp32", "small_192x136_8b_qp22", "small_128x128_10b_qp37", "wpp_256x192_8b_qp27"]


def load_case(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k: int(g[k]) for k in ("width", "height", "bit_depth", "frames", "qp", "wpp", "seed")}
    frames = [(g[f"ctus{i}"], g[f"rec{i}"]) for i in range(cfg["frames"])]
    return cfg, frames


def split_rec(rec, w, h):
    ny = w * h
    return [rec[:ny].reshape(h, w), rec[ny:ny + ny // 4].reshape(h // 2, w // 2), rec[ny + ny // 4:].reshape(h // 2, w // 2)]


def _z2xy():
    xs, ys = np.zeros(256, int), np.zeros(256, int)
    for z in range(256):
        for b in range(4):
            xs[z] |= ((z >> (2 * b)) & 1) << b
            ys[z] |= ((z >> (2 * b + 1)) & 1) << b
    return xs, ys


def inside_mask(n_ctus, width, height):
    """[ctu, z] -> True when the 4x4 partition lies inside the picture"""
    xs, ys = _z2xy()
    wc = (width + 63) // 64
    m = np.zeros((n_ctus, 256), bool)
    for a in range(n_ctus):
        m[a] = ((a % wc) * 64 + xs * 4 < width) & ((a // wc) * 64 + ys * 4 < height)
    return m


def assert_ctus_equal(got, want, what, size=None):
    """bit-exact comparison of per-CTU results (decisions, costs, coefficients).

    `size` = (width, height) when `want` comes from a dump of the real reference: that dump is taken after
    the whole picture was encoded, i.e. after TComPic::compressMotion (TEncGOP.cpp:1660) has overwritten
    m_pePredMode of every 16x16 block with its first partition's value.  Inside the picture every partition
    of an I slice is MODE_INTRA either way; outside it compressSlice leaves NUMBER_OF_PREDICTION_MODES, so
    pred_mode is compared inside the picture only.  Every other array is compared everywhere."""
    assert len(got) == len(want)
    if size is not None:
        m = inside_mask(len(got), *size)
        assert np.array_equal(got["pred_mode"][m], want["pred_mode"][m]), f"{what}: pred_mode differs inside the picture"
        got = got.copy(); got["pred_mode"] = want["pred_mode"]
    for f in ("total_bits", "total_dist"):
        assert np.array_equal(got[f], want[f]), f"{what}: {f} differs at CTU {np.nonzero(got[f] != want[f])[0][:5]}"
    assert np.array_equal(got["total_cost"], want["total_cost"]), f"{what}: total_cost differs"
    for f in META_FIELDS:
        if not np.array_equal(got[f], want[f]):
            bad = np.nonzero((got[f] != want[f]).reshape(len(got), -1).any(axis=1))[0]
            raise AssertionError(f"{what}: {f} differs in CTUs {bad[:8]}")
    for f in ("coeff_y", "coeff_cb", "coeff_cr"):
        assert np.array_equal(got[f], want[f]), f"{what}: {f} differs"


def assert_rec_equal(got_planes, want_flat, w, h, what):
    want = split_rec(want_flat, w, h)
    for k in range(3):
        assert np.array_equal(got_planes[k], want[k]), f"{what}: reconstruction plane {k} differs at {np.count_nonzero(got_planes[k] != want[k])} samples"


LDP_CASES = ["ldp_192x128_8b_qp32", "ldp_200x136_8b_qp24", "ldpwpp_256x136_8b_qp30"]      # P slices (encoder_lowdelay_P_main.cfg)
B_CASES = ["ra_192x128_10b_qp32", "ldb_200x136_8b_qp30"]       # B slices: encoder_randomaccess_main10.cfg, encoder_lowdelay_main.cfg
_S_KEYS = ("poc", "slice_type", "qp", "lambda", "sqrt_lambda", "weight_cb", "weight_cr", "lambda_motion_sad", "lambda_motion_sse",
           "col_from_l0", "col_ref_idx", "tmvp", "mvd_l1_zero", "max_merge_cand", "check_ldc", "cabac_init_type")


def load_ldp_case(name, records=None, sao=None, bits=None):
    """-> (cfg dict, list of 'S' records, {poc: 'F' record}) in the layout of tests/hmd2.py; `records` (a list) receives
    every record in stream order, `sao` (a dict) the SAO decisions by POC, `bits` (a dict) the encodeSlice output by POC"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k: int(g[k]) for k in ("width", "height", "bit_depth", "frames", "seed")}
    cfg["wpp"] = int(g["wpp"]) if "wpp" in g else 0
    slices, finals = [], {}
    pending_q = None
    pending_l = None
    for i in range(int(g["num_records"])):
        if chr(int(g[f"r{i}_tag"])) == "L":                      # LCU-level rate control: the lambda of every CTU of the slice that follows (tests/hmd2.py 'L' record)
            pending_l = {"tag": "L", "ctu_lambda": g[f"r{i}_ctu_lambda"], "ctu_qp": g[f"r{i}_ctu_qp"]}
            if records is not None:
                records.append(pending_l)
            continue
        if chr(int(g[f"r{i}_tag"])) == "Q":                      # cu_qp_delta side data of the slice that follows (tests/hmd2.py 'Q' record)
            pending_q = {"tag": "Q", "qp": g[f"r{i}_qp"], "activity": g[f"r{i}_activity"]}
            for k in ("max_cu_dqp_depth", "dqp_flag_in", "dqp_flag_out", "aq_range", "avg_activity"):
                pending_q[k] = g[f"r{i}_{k}"][()]
            if records is not None:
                records.append(pending_q)
            continue
        if chr(int(g[f"r{i}_tag"])) == "A":                      # SAO decisions of the picture (tests/hmd2.py 'A' record)
            r = {"tag": "A", "poc": int(g[f"r{i}_poc"]), "depth": int(g[f"r{i}_depth"]), "enabled": tuple(int(v) for v in g[f"r{i}_enabled"]), "sao": g[f"r{i}_sao"]}
            if sao is not None:
                sao[r["poc"]] = r
            if records is not None:
                records.append(r)
            continue
        if chr(int(g[f"r{i}_tag"])) == "B":                      # slice data bytes of the picture (tests/hmd2.py 'B' record)
            sizes, data = g[f"r{i}_sub_sizes"], g[f"r{i}_sub_bytes"].tobytes()
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
            r = {"tag": "B", "poc": int(g[f"r{i}_poc"]), "substreams": [data[offs[k]:offs[k + 1]] for k in range(len(sizes))],
                 "next_cabac_init_type": int(g[f"r{i}_next_cabac_init_type"]), "num_bins": int(g[f"r{i}_num_bins"])}
            if bits is not None:
                bits[r["poc"]] = r
            if records is not None:
                records.append(r)
            continue
        r = {"tag": chr(int(g[f"r{i}_tag"])), "num_ref_idx": tuple(int(v) for v in g[f"r{i}_num_ref_idx"]),
             "ref_poc": g[f"r{i}_ref_poc"], "ref_long_term": g[f"r{i}_ref_long_term"], "rec": [g[f"r{i}_rec{c}"] for c in range(3)]}
        if r["tag"] == "S":
            for k in _S_KEYS:
                r[k] = g[f"r{i}_{k}"][()]
            r["ctus"] = g[f"r{i}_ctus"]
            r["dqp"], pending_q = pending_q, None               # None unless cu_qp_delta was enabled
            r["lcu_rc"], pending_l = pending_l, None            # None unless the LCU-level rate control ran: its lambda and QP per CTU
            slices.append(r)
        else:
            r["poc"] = int(g[f"r{i}_poc"]); r["slice_type"] = int(g[f"r{i}_slice_type"]); r["motion"] = g[f"r{i}_motion"]
            finals[r["poc"]] = r
        if records is not None:
            records.append(r)
    return cfg, slices, finals


DQP_CASES = ["aq_i_256x192_8b_qp30", "aq_iwpp_320x200_10b_qp27", "aq_ldp_256x136_8b_qp32", "aq_ra_192x128_10b_qp30", "rc_ldp_256x128_8b"]
LCU_RC_CASES = ["rc2_ldp_256x128_8b", "rc2_i_256x192_10b", "rc2_ra_192x128_10b"]   # n4 stage 2, oracle only so far: the LCU-level rate control hands every CTU a QP and a lambda


INTER_PAIRS = [("skip", "skip"), ("merge_flag", "merge_flag"), ("merge_idx", "merge_idx"), ("inter_dir", "inter_dir")]


def assert_inter_ctus_equal(ctus, ictus, want, what):
    """bit-exact comparison of a P slice: everything assert_ctus_equal checks plus the motion data"""
    for f in ("total_bits", "total_dist", "total_cost", "depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx",
              "cbf", "tskip", "coeff_y", "coeff_cb", "coeff_cr"):
        assert np.array_equal(ctus[f], want[f]), f"{what}: {f} differs in CTUs {np.nonzero((ctus[f] != want[f]).reshape(len(ctus), -1).any(axis=1))[0][:8]}"
    for f, g in INTER_PAIRS:
        assert np.array_equal(ictus[f], want[g]), f"{what}: {f} differs"
    for l in range(2):
        for f, g in (("mv", "mv%d"), ("mvd", "mvd%d"), ("ref_idx", "ref_idx%d"), ("mvp_idx", "mvp_idx%d"), ("mvp_num", "mvp_num%d")):
            assert np.array_equal(ictus[f][:, l], want[g % l]), f"{what}: {g % l} differs"


def ldp_slice_inputs(r, finals):
    """(slice_params, ref_pics) in the form hm355.Encoder.compress_inter takes, from an 'S' record and the 'F' records"""
    sp = {k: r[k] for k in ("slice_type", "qp", "lambda", "poc", "cabac_init_type", "num_ref_idx", "ref_poc", "col_from_l0", "col_ref_idx", "tmvp",
                            "mvd_l1_zero", "max_merge_cand", "check_ldc", "lambda_motion_sad", "lambda_motion_sse")}
    sp["chroma_weight"] = r["weight_cb"]
    refs = {}
    for poc in set(int(r["ref_poc"][l][i]) for l in range(2) for i in range(r["num_ref_idx"][l])):
        f = finals[poc]; m = f["motion"]
        refs[poc] = dict(slice_type=f["slice_type"], rec=f["rec"], pred_mode=m["pred_mode"], mv=[m["mv0"], m["mv1"]],
                         ref_idx=[m["ref_idx0"], m["ref_idx1"]], num_ref_idx=f["num_ref_idx"], ref_poc=f["ref_poc"],
                         ref_long_term=f["ref_long_term"])
    return sp, refs


LDP_LONG_CASES = ["ldp2gop_256x128_8b_qp34"]     # I + two low-delay GOPs: reference picture sets beyond the first GOP, a P slice on the B context table
DBK_CASES = ["dbk_ldp_200x136_8b_qp30", "dbk_ldb_192x128_10b_qp34"]    # SAO off: 'F' record = deblocked 'S' record


def split_fixture_ctus(want):
    """fixture CTU records (tests/hmd2.py CTU_DT) -> (ctus, ictus) arrays in the C-ABI / oracle layouts"""
    import hm355
    n = len(want)
    ctus, ictus = np.zeros(n, hm355.CTU_DTYPE), np.zeros(n, hm355.CTU_INTER_DTYPE)
    for f in ("total_cost", "total_bits", "total_dist", "depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx",
              "cbf", "tskip", "coeff_y", "coeff_cb", "coeff_cr"):
        ctus[f] = want[f]
    for f, g in INTER_PAIRS:
        ictus[f] = want[g]
    for l in range(2):
        for f, g in (("mv", "mv%d"), ("mvd", "mvd%d"), ("ref_idx", "ref_idx%d"), ("mvp_idx", "mvp_idx%d"), ("mvp_num", "mvp_num%d")):
            ictus[f][:, l] = want[g % l]
    return ctus, ictus


def normalise_sao(params):
    """SAO block parameters (n, 3, 35) reduced to the fields the mode defines: OFF -> nothing, MERGE -> direction, NEW -> type, band position, offsets"""
    p = np.array(params, np.int32).copy()
    off, mrg = p[:, :, 0] == 0, p[:, :, 0] == 2
    p[off, 1:] = 0
    p[mrg, 2:] = 0
    return p


def read_mirror_bits(path, frames):
    """the .bits side file of hm355_encmain: per picture u32 numSubstreams, then per substream u32 size + bytes -> [[bytes]]"""
    import struct
    buf = open(path, "rb").read()
    off, out = 0, []
    for _ in range(frames):
        n, = struct.unpack_from("<I", buf, off); off += 4
        subs = []
        for _ in range(n):
            nb, = struct.unpack_from("<I", buf, off); off += 4
            subs.append(buf[off:off + nb]); off += nb
        out.append(subs)
    assert off == len(buf)
    return out


YUVIO_CASES = ["yuvio_100x60_8to10_pad4x4_out8", "yuvio_72x40_10to10_out10", "yuvio_90x50_10to8_pad6x6_out10", "yuvio_64x64_8to8_out8",
               "yuvio_130x70_8to10_pad6x2_out10"]


def load_yuvio_case(name):
    """-> dict: geometry / bit depths, per-frame raw input bytes, per-frame planes as TVideoIOYuv::read left them, per-frame bytes TVideoIOYuv::write produced"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    c = {k: int(g[k]) for k in ("file_w", "file_h", "file_bd", "internal_bd", "pad_x", "pad_y", "out_bd", "frames")}
    w, h = c["file_w"] + c["pad_x"], c["file_h"] + c["pad_y"]
    fb = c["file_w"] * c["file_h"] * 3 // 2 * (2 if c["file_bd"] > 8 else 1)
    ob = c["file_w"] * c["file_h"] * 3 // 2 * (2 if c["out_bd"] > 8 else 1)
    pn = w * h * 3 // 2
    raw, out = g["raw"].tobytes(), g["out"].tobytes()
    c["raw"] = [raw[i * fb:(i + 1) * fb] for i in range(c["frames"])]
    c["out"] = [out[i * ob:(i + 1) * ob] for i in range(c["frames"])]
    c["planes"] = []
    for i in range(c["frames"]):
        p = g["planes"][i * pn:(i + 1) * pn]
        c["planes"].append([p[:w * h].reshape(h, w), p[w * h:w * h * 5 // 4].reshape(h // 2, w // 2), p[w * h * 5 // 4:].reshape(h // 2, w // 2)])
    c["width"], c["height"] = w, h
    return c


# ---- full-size pins (tests/gen_golden_full.py): digests of what the reference left, compared CTU by CTU ----
FULL_CASES = ["full_c4_3840x2160_10b_wpp_qp32", "full_c2_1920x1080_10b_qp32", "full_c3_ldp_1920x1080_8b_wpp_qp32", "full_c5_ra_3840x2160_10b_wpp_qp32"]
_DIGEST_CTU = ("total_cost", "total_bits", "total_dist", "depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx", "cbf", "tskip",
               "coeff_y", "coeff_cb", "coeff_cr")
_DIGEST_INTER = ("skip", "merge_flag", "merge_idx", "inter_dir", "mv", "mvd", "ref_idx", "mvp_idx", "mvp_num")


def ctu_digests(ctus, ictus=None):
    """(n, 20) uint8: SHA-1 per CTU over its decision arrays, costs and coefficients (C-ABI layout, fixed field order), and its motion
    data when `ictus` is given"""
    import hashlib
    out = np.zeros((len(ctus), 20), np.uint8)
    for a in range(len(ctus)):
        h = hashlib.sha1()
        for f in _DIGEST_CTU:
            h.update(np.ascontiguousarray(ctus[f][a]).tobytes())
        if ictus is not None:
            for f in _DIGEST_INTER:
                h.update(np.ascontiguousarray(ictus[f][a]).tobytes())
        out[a] = np.frombuffer(h.digest(), np.uint8)
    return out


def md5_of(arr):
    import hashlib
    return np.frombuffer(hashlib.md5(np.ascontiguousarray(arr).tobytes()).digest(), np.uint8).copy()


def load_full_case(name):
    """-> (cfg, [picture dict in coding order]) of a full-size pin; every picture: the slice parameters compressSlice saw, per-CTU digests,
    MD5s of the pre-deblocking and the finished planes, SAO flags + digest, per-substream MD5s"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k: int(g[k]) for k in ("width", "height", "bit_depth", "frames", "seed", "wpp", "pictures")}
    pics = []
    for i in range(cfg["pictures"]):
        p = {k: g[f"p{i}_{k}"][()] for k in _S_KEYS}
        for k in ("num_ref_idx", "ref_poc", "ref_long_term", "ctu_sha1", "rec_md5", "final_md5", "sao_enabled", "sao_md5", "sao_depth", "sub_sizes", "sub_md5",
                  "next_cabac_init_type", "num_bins"):
            p[k] = g[f"p{i}_{k}"]
        p["num_ref_idx"] = tuple(int(v) for v in p["num_ref_idx"])
        pics.append(p)
    return cfg, pics
