"""Regenerates tests/golden/*.npz from the REAL reference encoder (oracle/_ref/hm_dump, built from
/root/reference by oracle/Makefile.ref).  Runs only in the development container: the GPU box has no
reference.  The fixtures are data (inputs are regenerated from the seeded generator, expected outputs
are stored); no reference source text is stored.

    python tests/gen_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hm-16.2_amd"))
import synth  # noqa: E402

HM_DUMP = os.path.join(ROOT, "oracle", "_ref", "hm_dump")
REF_CFG = "/root/reference/cfg"
GOLD = os.path.join(ROOT, "tests", "golden")

# name, width, height, bit depth, frames, qp, wpp, seed
CASES = [
    ("c1_416x240_8b_qp32", 416, 240, 8, 1, 32, 0, 1234),       # BASELINE configs[0] geometry (plumbing case)
    ("wpp_416x240_10b_qp32", 416, 240, 10, 1, 32, 1, 1234),    # main10 + WaveFrontSynchro
    ("small_192x136_8b_qp22", 192, 136, 8, 1, 22, 0, 4321),    # ragged bottom edge (8-pixel CU row), low QP
    ("small_128x128_10b_qp37", 128, 128, 10, 2, 37, 0, 99),    # two pictures, high QP
    ("wpp_256x192_8b_qp27", 256, 192, 8, 1, 27, 1, 7),
]

CTU_DTYPE = np.dtype([("total_cost", "<f8"), ("total_bits", "<u4"), ("total_dist", "<u4"),
                      ("depth", "u1", 256), ("part_size", "u1", 256), ("pred_mode", "u1", 256),
                      ("intra_dir_luma", "u1", 256), ("intra_dir_chroma", "u1", 256), ("tr_idx", "u1", 256),
                      ("cbf", "u1", (3, 256)), ("tskip", "u1", (3, 256)),
                      ("coeff_y", "<i4", 4096), ("coeff_cb", "<i4", 1024), ("coeff_cr", "<i4", 1024)])


def parse_dump(path):
    d = open(path, "rb").read()
    assert d[:4] == b"HMD1"
    w, h, bd, ctu, nf = struct.unpack("<5I", d[4:24])
    off = 24
    frames = []
    for _ in range(nf):
        poc, n = struct.unpack("<2I", d[off:off + 8])
        off += 8
        ctus = np.frombuffer(d, dtype=CTU_DTYPE, count=n, offset=off).copy()
        off += n * CTU_DTYPE.itemsize
        ny = w * h
        rec = np.frombuffer(d, dtype="<u2", count=ny * 3 // 2, offset=off).copy()
        off += ny * 3
        frames.append((ctus, rec))
    return frames


def run_case(name, w, h, bd, nf, qp, wpp, seed):
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "in.yuv")
        synth.write_yuv(yuv, w, h, bd, nf, seed)
        cfg = os.path.join(REF_CFG, "encoder_intra_main10.cfg" if bd == 10 else "encoder_intra_main.cfg")
        dump = os.path.join(td, "dump.bin")
        cmd = [HM_DUMP, "enc", "-c", cfg, "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "50", "-f", str(nf),
               f"--InputBitDepth={bd}", "-q", str(qp), "-b", os.path.join(td, "o.bin"), "-o", os.path.join(td, "r.yuv"),
               # deblocking/SAO run after compressSlice; switched off so the dumped picture is compressSlice's output
               "--DeblockingFilterControlPresent=1", "--LoopFilterDisable=1", "--SAO=0",
               f"--WaveFrontSynchro={wpp}", "--", dump]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        frames = parse_dump(dump)
    out = {"width": w, "height": h, "bit_depth": bd, "frames": nf, "qp": qp, "wpp": wpp, "seed": seed}
    for i, (ctus, rec) in enumerate(frames):
        out[f"ctus{i}"] = ctus
        out[f"rec{i}"] = rec
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "ok", sum(int(c["total_bits"].sum()) for c, _ in frames), "bits")


def gen_kat():
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "kat.bin")
        subprocess.run([HM_DUMP, "kat", p, "20261003"], check=True)
        d = open(p, "rb").read()
    assert d[:4] == b"KAT1"
    off = 4
    recs = {"dist_in": [], "dist_out": [], "tr_in": [], "tr_out": []}
    while off < len(d):
        tag, nin = struct.unpack("<2I", d[off:off + 8]); off += 8
        vin = np.frombuffer(d, "<i4", nin, off); off += 4 * nin
        nout, = struct.unpack("<I", d[off:off + 4]); off += 4
        vout = np.frombuffer(d, "<i4", nout, off); off += 4 * nout
        if tag in (1, 2, 3):
            recs["dist_in"].append(np.concatenate([[tag], vin]).astype(np.int32))
            recs["dist_out"].append(int(vout[0]))
        else:
            recs["tr_in"].append(np.concatenate([[tag], vin]).astype(np.int32))
            recs["tr_out"].append(vout.astype(np.int32))
    def flat(lst):
        off = np.cumsum([0] + [len(x) for x in lst]).astype(np.int64)
        return np.concatenate(lst).astype(np.int32), off
    di, dio = flat(recs["dist_in"]); ti, tio = flat(recs["tr_in"]); to, too = flat(recs["tr_out"])
    np.savez_compressed(os.path.join(GOLD, "kat_primitives.npz"), dist_in=di, dist_in_off=dio,
                        dist_out=np.array(recs["dist_out"], np.int64), tr_in=ti, tr_in_off=tio, tr_out=to, tr_out_off=too)
    print("kat ok", len(recs["dist_out"]), "distortion records,", len(recs["tr_out"]), "transform records")


# inter (low-delay P) cases: name, width, height, bit depth, frames, qp, seed.  encoder_lowdelay_P_main.cfg as is (4 references,
# deblocking + SAO on): the reference pictures a P slice sees are inputs of compressSlice, so they are part of the fixture.
LDP_CASES = [
    ("ldp_192x128_8b_qp32", 192, 128, 8, 6, 32, 1234),
    ("ldp_200x136_8b_qp24", 200, 136, 8, 5, 24, 5),           # picture not a multiple of the CTU size
    ("ldpwpp_256x136_8b_qp30", 256, 136, 8, 4, 30, 77, 1),    # WaveFrontSynchro=1, last CTU row partial (2Nx2N integer-MV carry)
    # B slices: random access (hierarchical GOP 8, both directions) and low-delay B (list 1 == list 0, mvd_l1_zero)
    ("ra_192x128_10b_qp32", 192, 128, 10, 9, 32, 4321, 0, "encoder_randomaccess_main10.cfg"),
    ("ldb_200x136_8b_qp30", 200, 136, 8, 4, 30, 99, 0, "encoder_lowdelay_main.cfg"),
    # two full low-delay GOPs after the I picture (pins the C++ host mirror's reference picture sets beyond the first GOP)
    ("ldp2gop_256x128_8b_qp34", 256, 128, 8, 9, 34, 2024),
    # deblocking pins: SAO off, so the finished picture ('F' record) is the deblocked pre-deblock reconstruction of the 'S' record
    ("dbk_ldp_200x136_8b_qp30", 200, 136, 8, 3, 30, 31, 0, "encoder_lowdelay_P_main.cfg", ("--SAO=0",)),
    ("dbk_ldb_192x128_10b_qp34", 192, 128, 10, 3, 34, 32, 0, "encoder_lowdelay_main10.cfg", ("--SAO=0",)),
    # cu_qp_delta.  AdaptiveQP (TEncPreanalyzer activity -> per-CTU QP, MaxCuDQPDepth 0) on I / P / B clips, and the
    # picture-level rate control (cu_qp_delta enabled with every CTU at the slice QP the rate model chose)
    ("aq_i_256x192_8b_qp30", 256, 192, 8, 2, 30, 51, 0, "encoder_intra_main.cfg", ("--AdaptiveQP=1",)),
    ("aq_iwpp_320x200_10b_qp27", 320, 200, 10, 2, 27, 52, 1, "encoder_intra_main10.cfg", ("--AdaptiveQP=1", "--MaxQPAdaptationRange=8")),
    ("aq_ldp_256x136_8b_qp32", 256, 136, 8, 4, 32, 53, 1, "encoder_lowdelay_P_main.cfg", ("--AdaptiveQP=1",)),
    ("aq_ra_192x128_10b_qp30", 192, 128, 10, 5, 30, 54, 0, "encoder_randomaccess_main10.cfg", ("--AdaptiveQP=1",)),
    ("rc_ldp_256x128_8b", 256, 128, 8, 5, 32, 55, 1, "encoder_lowdelay_P_main.cfg", ("--RateControl=1", "--TargetBitrate=400000", "--LCULevelRateControl=0", "--InitialQP=30")),
    # n4 stage 2: the LCU-level rate model gives every CTU of the P pictures its own QP AND lambda ('L' record)
    ("rc2_ldp_256x128_8b", 256, 128, 8, 5, 32, 56, 1, "encoder_lowdelay_P_main.cfg", ("--RateControl=1", "--TargetBitrate=400000", "--LCULevelRateControl=1", "--InitialQP=30")),
    ("rc2_i_256x192_10b", 256, 192, 10, 3, 32, 57, 0, "encoder_intra_main10.cfg", ("--RateControl=1", "--TargetBitrate=3000000", "--LCULevelRateControl=1", "--InitialQP=28")),
    ("rc2_ra_192x128_10b", 192, 128, 10, 5, 32, 58, 0, "encoder_randomaccess_main10.cfg", ("--RateControl=1", "--TargetBitrate=300000", "--LCULevelRateControl=1", "--InitialQP=30")),
]
S_KEYS = ("poc", "slice_type", "qp", "lambda", "sqrt_lambda", "weight_cb", "weight_cr", "lambda_motion_sad", "lambda_motion_sse",
          "col_from_l0", "col_ref_idx", "tmvp", "mvd_l1_zero", "max_merge_cand", "check_ldc", "cabac_init_type")


def run_ldp_case(name, w, h, bd, nf, qp, seed, wpp=0, cfg="encoder_lowdelay_P_main.cfg", extra=()):
    import hmd2
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "in.yuv")
        synth.write_yuv(yuv, w, h, bd, nf, seed)
        dump = os.path.join(td, "dump2.bin")
        cmd = [HM_DUMP, "enc2", "-c", os.path.join(REF_CFG, cfg), "-i", yuv, "-wdt", str(w), "-hgt", str(h),
               "-fr", "50", "-f", str(nf), f"--InputBitDepth={bd}", "-q", str(qp), "-b", os.path.join(td, "o.bin"),
               "-o", os.path.join(td, "r.yuv")] + (["--WaveFrontSynchro=1"] if wpp else []) + list(extra) + ["--", dump]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        recs = hmd2.parse(dump, w, h)
    out = {"width": w, "height": h, "bit_depth": bd, "frames": nf, "seed": seed, "wpp": wpp, "num_records": len(recs)}
    for i, r in enumerate(recs):
        out[f"r{i}_tag"] = np.array(ord(r["tag"]))
        if r["tag"] == "L":
            out[f"r{i}_ctu_lambda"] = r["ctu_lambda"]; out[f"r{i}_ctu_qp"] = r["ctu_qp"]
            continue
        if r["tag"] == "Q":
            for k in ("max_cu_dqp_depth", "dqp_flag_in", "dqp_flag_out", "aq_range", "avg_activity"):
                out[f"r{i}_{k}"] = np.array(r[k])
            out[f"r{i}_qp"] = r["qp"]; out[f"r{i}_activity"] = r["activity"]
            continue
        if r["tag"] == "A":
            out[f"r{i}_poc"] = np.array(r["poc"]); out[f"r{i}_depth"] = np.array(r["depth"])
            out[f"r{i}_enabled"] = np.array(r["enabled"]); out[f"r{i}_sao"] = r["sao"]
            continue
        if r["tag"] == "B":
            out[f"r{i}_poc"] = np.array(r["poc"]); out[f"r{i}_sub_sizes"] = np.array([len(b) for b in r["substreams"]], np.uint32)
            out[f"r{i}_sub_bytes"] = np.frombuffer(b"".join(r["substreams"]), np.uint8).copy()
            out[f"r{i}_next_cabac_init_type"] = np.array(r["next_cabac_init_type"]); out[f"r{i}_num_bins"] = np.array(r["num_bins"])
            continue
        out[f"r{i}_num_ref_idx"] = np.array(r["num_ref_idx"]); out[f"r{i}_ref_poc"] = r["ref_poc"]; out[f"r{i}_ref_long_term"] = r["ref_long_term"]
        for c in range(3):
            out[f"r{i}_rec{c}"] = r["rec"][c]
        if r["tag"] == "S":
            for k in S_KEYS:
                out[f"r{i}_{k}"] = np.array(r[k])
            out[f"r{i}_ctus"] = r["ctus"]
        else:
            out[f"r{i}_poc"] = np.array(r["poc"]); out[f"r{i}_slice_type"] = np.array(r["slice_type"]); out[f"r{i}_motion"] = r["motion"]
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "ok", len(recs), "records")


# picture ingest / output cases: name, file width, height, file bit depth, internal bit depth, pad x, pad y, output bit depth, seed.
# Random samples over the whole range of the file bit depth, so that down-conversions round and clip.
YUVIO_CASES = [
    ("yuvio_100x60_8to10_pad4x4_out8", 100, 60, 8, 10, 4, 4, 8, 1),
    ("yuvio_72x40_10to10_out10", 72, 40, 10, 10, 0, 0, 10, 2),
    ("yuvio_90x50_10to8_pad6x6_out10", 90, 50, 10, 8, 6, 6, 10, 3),
    ("yuvio_64x64_8to8_out8", 64, 64, 8, 8, 0, 0, 8, 4),
    ("yuvio_130x70_8to10_pad6x2_out10", 130, 70, 8, 10, 6, 2, 10, 5),
]


def gen_yuvio():
    nf = 2
    for name, fw, fh, fbd, ibd, px, py, obd, seed in YUVIO_CASES:
        rng = np.random.default_rng(seed)
        n = fw * fh * 3 // 2 * nf
        raw = (rng.integers(0, 1 << fbd, n).astype(np.uint16).astype("<u2").tobytes() if fbd > 8 else rng.integers(0, 256, n).astype(np.uint8).tobytes())
        with tempfile.TemporaryDirectory() as td:
            fi, fo, fd = os.path.join(td, "in.yuv"), os.path.join(td, "out.yuv"), os.path.join(td, "dump.bin")
            open(fi, "wb").write(raw)
            subprocess.run([HM_DUMP, "yuvio", fi, str(fw), str(fh), str(fbd), str(ibd), str(px), str(py), str(nf), str(obd), fo, fd], check=True)
            planes = np.frombuffer(open(fd, "rb").read(), "<u2").copy()
            out = np.frombuffer(open(fo, "rb").read(), np.uint8).copy()
        assert len(planes) == (fw + px) * (fh + py) * 3 // 2 * nf
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), file_w=fw, file_h=fh, file_bd=fbd, internal_bd=ibd, pad_x=px, pad_y=py, out_bd=obd, frames=nf,
                            raw=np.frombuffer(raw, np.uint8), planes=planes, out=out)
        print(name, "ok")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if "--yuvio-only" in sys.argv:
        gen_yuvio(); sys.exit(0)
    if "--ldp-only" not in sys.argv:
        gen_kat(); gen_yuvio()
    if "--kat-only" not in sys.argv:
        if "--ldp-only" not in sys.argv:
            for c in CASES:
                run_case(*c)
        for c in LDP_CASES:
            if "--only" in sys.argv and c[0] != sys.argv[sys.argv.index("--only") + 1]:
                continue
            run_ldp_case(*c)
