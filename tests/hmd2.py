"""Reader of the HMD2 record stream written by `oracle/_ref/hm_dump enc2` (oracle/ref_harness.cpp):
'S' records = one slice as TEncSlice::compressSlice saw and left it, 'F' records = the finished picture
as later pictures reference it (final reconstruction + compressed motion field), 'A' records = the SAO decisions of the picture,
'B' records = the substream bytes TEncSlice::encodeSlice wrote for the slice."""
import struct

import numpy as np

CTU_DT = np.dtype([("total_cost", "<f8"), ("total_bits", "<u4"), ("total_dist", "<u4"),
                   ("depth", "u1", 256), ("part_size", "u1", 256), ("pred_mode", "u1", 256),
                   ("intra_dir_luma", "u1", 256), ("intra_dir_chroma", "u1", 256), ("tr_idx", "u1", 256),
                   ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
                   ("skip", "u1", 256), ("merge_flag", "u1", 256), ("merge_idx", "u1", 256), ("inter_dir", "u1", 256),
                   ("mv0", "<i2", (256, 2)), ("mvd0", "<i2", (256, 2)), ("ref_idx0", "i1", 256), ("mvp_idx0", "i1", 256), ("mvp_num0", "i1", 256),
                   ("mv1", "<i2", (256, 2)), ("mvd1", "<i2", (256, 2)), ("ref_idx1", "i1", 256), ("mvp_idx1", "i1", 256), ("mvp_num1", "i1", 256),
                   ("coeff_y", "<i4", 4096), ("coeff_cb", "<i4", 1024), ("coeff_cr", "<i4", 1024)])
MOT_DT = np.dtype([("pred_mode", "u1", 256), ("mv0", "<i2", (256, 2)), ("ref_idx0", "i1", 256), ("mv1", "<i2", (256, 2)), ("ref_idx1", "i1", 256)])


def _planes(buf, off, w, h):
    out = []
    for c in range(3):
        cw, ch = (w, h) if c == 0 else (w // 2, h // 2)
        out.append(np.frombuffer(buf, "<u2", cw * ch, off).reshape(ch, cw).copy())
        off += 2 * cw * ch
    return out, off


def _ref_lists(buf, off):
    n = struct.unpack_from("<2i", buf, off); off += 8
    poc = np.frombuffer(buf, "<i4", 32, off).reshape(2, 16).copy(); off += 128
    lt = np.frombuffer(buf, "<i4", 32, off).reshape(2, 16).copy(); off += 128
    return {"num_ref_idx": n, "ref_poc": poc, "ref_long_term": lt}, off


def parse(path, width, height):
    buf = open(path, "rb").read()
    assert buf[:4] == b"HMD2"
    off, recs = 4, []
    while off < len(buf):
        tag = buf[off:off + 1]; off += 1
        if tag == b"S":
            r = {"tag": "S"}
            r["poc"], r["slice_type"], r["qp"], r["tlayer"], r["depth"] = struct.unpack_from("<5i", buf, off); off += 20
            r["lambda"], r["sqrt_lambda"], r["weight_cb"], r["weight_cr"] = struct.unpack_from("<4d", buf, off); off += 32
            r["lambda_motion_sad"], r["lambda_motion_sse"] = struct.unpack_from("<2I", buf, off); off += 8
            rl, off = _ref_lists(buf, off); r.update(rl)
            (r["col_from_l0"], r["col_ref_idx"], r["tmvp"], r["mvd_l1_zero"], r["max_merge_cand"], r["check_ldc"], r["cabac_init_type"]) = struct.unpack_from("<7i", buf, off); off += 28
            r["l1_to_l0"] = np.frombuffer(buf, "<i4", 16, off).copy(); off += 64
            n, = struct.unpack_from("<I", buf, off); off += 4
            r["ctus"] = np.frombuffer(buf, CTU_DT, n, off).copy(); off += n * CTU_DT.itemsize
            r["rec"], off = _planes(buf, off, width, height)
        elif tag == b"F":
            r = {"tag": "F"}
            r["poc"], = struct.unpack_from("<i", buf, off); off += 4
            r["rec"], off = _planes(buf, off, width, height)
            r["slice_type"], = struct.unpack_from("<i", buf, off); off += 4
            rl, off = _ref_lists(buf, off); r.update(rl)
            n, = struct.unpack_from("<I", buf, off); off += 4
            r["motion"] = np.frombuffer(buf, MOT_DT, n, off).copy(); off += n * MOT_DT.itemsize
        elif tag == b"Q":                                   # cu_qp_delta side data of the slice whose 'S' record follows
            r = {"tag": "Q"}
            r["max_cu_dqp_depth"], r["dqp_flag_in"], r["dqp_flag_out"], r["aq_range"], n = struct.unpack_from("<4iI", buf, off); off += 20
            r["qp"] = np.frombuffer(buf, "i1", n * 256, off).reshape(n, 256).copy(); off += n * 256
            nu, = struct.unpack_from("<I", buf, off); off += 4
            r["avg_activity"], r["activity"] = 0.0, np.zeros(0)
            if nu:
                r["avg_activity"], = struct.unpack_from("<d", buf, off); off += 8
                r["activity"] = np.frombuffer(buf, "<f8", nu, off).copy(); off += 8 * nu
        elif tag == b"L":                                   # LCU-level rate control: the lambda of every CTU's search (slice whose 'S' record follows)
            r = {"tag": "L"}
            n, = struct.unpack_from("<I", buf, off); off += 4
            r["ctu_lambda"] = np.frombuffer(buf, "<f8", n, off).copy(); off += 8 * n
            r["ctu_qp"] = np.frombuffer(buf, "<i4", n, off).copy(); off += 4 * n
        elif tag == b"A":
            r = {"tag": "A"}
            r["poc"], r["depth"], en0, en1, n = struct.unpack_from("<4iI", buf, off); off += 20
            r["enabled"] = (en0, en1)
            r["sao"] = np.frombuffer(buf, "<i4", n * 3 * 35, off).reshape(n, 3, 35).copy(); off += n * 3 * 35 * 4
        elif tag == b"B":
            r = {"tag": "B"}
            r["poc"], n = struct.unpack_from("<iI", buf, off); off += 8
            r["substreams"] = []
            for _ in range(n):
                nb, = struct.unpack_from("<I", buf, off); off += 4
                r["substreams"].append(bytes(buf[off:off + nb])); off += nb
            r["next_cabac_init_type"], r["num_bins"] = struct.unpack_from("<iI", buf, off); off += 8
        else:
            raise ValueError(f"bad record tag {tag!r} at {off - 1}")
        recs.append(r)
    return recs


def write(path, recs, bits=False):
    """the inverse of parse(): records (dicts as parse() returns them; missing S fields are written as 0) -> HMD2 stream.
    Without `bits` only the 'S' and 'F' records the search replays are written; with it also every 'B' record, preceded by the 'A'
    record of the same picture (the SAO syntax is part of the slice data)."""
    def ref_lists(r):
        return struct.pack("<2i", *[int(v) for v in r["num_ref_idx"]]) + np.ascontiguousarray(r["ref_poc"], "<i4").tobytes() + \
            np.ascontiguousarray(r["ref_long_term"], "<i4").tobytes()
    with open(path, "wb") as f:
        f.write(b"HMD2")
        for r in recs:
            if r["tag"] in ("A", "Q", "L") or (r["tag"] == "B" and not bits):
                continue                                  # SAO decisions, slice data bytes: not part of what the search replays
            if r["tag"] == "B":
                for a in recs:
                    if a["tag"] == "A" and a["poc"] == r["poc"]:
                        f.write(b"A" + struct.pack("<4iI", int(a["poc"]), int(a["depth"]), int(a["enabled"][0]), int(a["enabled"][1]), len(a["sao"])))
                        f.write(np.ascontiguousarray(a["sao"], "<i4").tobytes())
                f.write(b"B" + struct.pack("<iI", int(r["poc"]), len(r["substreams"])))
                for sub in r["substreams"]:
                    f.write(struct.pack("<I", len(sub)) + sub)
                f.write(struct.pack("<iI", int(r["next_cabac_init_type"]), int(r["num_bins"])))
                continue
            f.write(r["tag"].encode())
            rec = b"".join(np.ascontiguousarray(p, "<u2").tobytes() for p in r["rec"])
            if r["tag"] == "S":
                f.write(struct.pack("<5i", int(r["poc"]), int(r["slice_type"]), int(r["qp"]), int(r.get("tlayer", 0)), int(r.get("depth", 0))))
                f.write(struct.pack("<4d", float(r["lambda"]), float(r["sqrt_lambda"]), float(r["weight_cb"]), float(r["weight_cr"])))
                f.write(struct.pack("<2I", int(r["lambda_motion_sad"]), int(r["lambda_motion_sse"])))
                f.write(ref_lists(r))
                f.write(struct.pack("<7i", *[int(r[k]) for k in ("col_from_l0", "col_ref_idx", "tmvp", "mvd_l1_zero", "max_merge_cand", "check_ldc", "cabac_init_type")]))
                f.write(np.ascontiguousarray(r.get("l1_to_l0", np.zeros(16)), "<i4").tobytes())
                f.write(struct.pack("<I", len(r["ctus"])))
                f.write(np.ascontiguousarray(r["ctus"], CTU_DT).tobytes())
                f.write(rec)
            else:
                f.write(struct.pack("<i", int(r["poc"])))
                f.write(rec)
                f.write(struct.pack("<i", int(r["slice_type"])))
                f.write(ref_lists(r))
                f.write(struct.pack("<I", len(r["motion"])))
                f.write(np.ascontiguousarray(r["motion"], MOT_DT).tobytes())
