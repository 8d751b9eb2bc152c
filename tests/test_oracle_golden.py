"""CPU suite: the oracle (oracle/hm_oracle.c) against the golden vectors produced by the real
reference encoder (tests/gen_golden.py -> oracle/_ref/hm_dump).  This is what pins the oracle."""
import numpy as np
import pytest

import common
import synth


@pytest.mark.parametrize("name", common.CASES)
def test_oracle_matches_reference_fixture(built, name):
    import oracle
    cfg, frames = common.load_case(name)
    for i, (ctus, rec) in enumerate(frames):
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], i, cfg["seed"])
        got_rec, got_ctus = oracle.compress(planes, cfg["bit_depth"], cfg["qp"], cfg["wpp"])
        common.assert_ctus_equal(got_ctus, ctus, f"{name} frame {i}", (cfg["width"], cfg["height"]))
        common.assert_rec_equal(got_rec, rec, cfg["width"], cfg["height"], f"{name} frame {i}")


def _kat():
    return np.load(common.GOLD + "/kat_primitives.npz")


def test_oracle_distortion_kats(built):
    """SAD (with and without FEN row sub-sampling), SSE, SATD vs TComRdCost of the reference"""
    import oracle
    k = _kat()
    off = k["dist_in_off"]
    for r in range(len(k["dist_out"])):
        v = k["dist_in"][off[r]:off[r + 1]]
        tag = int(v[0])
        if tag == 1:
            bd, n, sub = int(v[1]), int(v[2]), int(v[3]); data = v[4:]
        else:
            bd, n, sub = int(v[1]), int(v[2]), 0; data = v[3:]
        a = data[:n * n].astype(np.int16).reshape(n, n); b = data[n * n:].astype(np.int16).reshape(n, n)
        got = oracle.dist({1: 0, 2: 1, 3: 2}[tag], a, b, bd, sub)
        assert got == int(k["dist_out"][r]), f"record {r} tag {tag} n {n} bd {bd}"


def test_oracle_transform_kats(built):
    """forward / inverse 4..32-point DCT and the 4x4 DST vs xTrMxN / xITrMxN of the reference"""
    import oracle
    k = _kat()
    ioff, ooff = k["tr_in_off"], k["tr_out_off"]
    for r in range(len(ioff) - 1):
        v = k["tr_in"][ioff[r]:ioff[r + 1]]
        tag, bd, n, dst = int(v[0]), int(v[1]), int(v[2]), int(v[3])
        blk = v[4:].astype(np.int32).reshape(n, n)
        want = k["tr_out"][ooff[r]:ooff[r + 1]].reshape(n, n)
        got = oracle.transform(tag == 5, blk, bd, dst)
        assert np.array_equal(got, want), f"record {r} tag {tag} n {n} bd {bd} dst {dst}"


@pytest.mark.parametrize("name", common.LDP_CASES + common.B_CASES)
def test_oracle_matches_reference_p_and_b_slices(built, name):
    """encoder_lowdelay_P_main.cfg / encoder_randomaccess_main10.cfg / encoder_lowdelay_main.cfg: every P or B slice of the clip,
    with the reference pictures (final reconstruction + motion field) and slice parameters exactly as the reference's
    compressSlice saw them; decisions, motion, coefficients, costs and the pre-deblocking reconstruction must match bit for bit."""
    import oracle
    cfg, slices, finals = common.load_ldp_case(name)
    n_p = 0
    for r in slices:
        if int(r["slice_type"]) == 2:
            continue
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        rec, ctus, ictus = oracle.compress_inter(planes, cfg["bit_depth"], r, finals, wpp=cfg["wpp"])
        common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], f"{name} POC {int(r['poc'])}")
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{name} POC {int(r['poc'])}: reconstruction plane {c}"
        n_p += 1
    assert n_p >= 3


@pytest.mark.parametrize("name", common.DBK_CASES)
def test_oracle_deblocking_matches_reference(built, name):
    """TComLoopFilter::loopFilterPic: the reference run with SAO off leaves the deblocked picture as the finished picture; the oracle's
    deblocking of the pre-deblocking reconstruction (with the CU / TU / motion data of the same slice) must equal it (I, P and B slices)."""
    import oracle
    cfg, slices, finals = common.load_ldp_case(name)
    for r in slices:
        ctus, ictus = common.split_fixture_ctus(r["ctus"])
        got = oracle.deblock(r["rec"], cfg["bit_depth"], int(r["qp"]), int(r["slice_type"]), r["ref_poc"], ctus, ictus)
        want = finals[int(r["poc"])]["rec"]
        for c in range(3):
            assert np.array_equal(got[c], want[c]), f"{name} POC {int(r['poc'])}: deblocked plane {c} differs at {int((got[c] != want[c]).sum())} samples"
        assert any(not np.array_equal(r["rec"][c], want[c]) for c in range(3)), "the fixture does not exercise the filter"


@pytest.mark.parametrize("name", common.LDP_CASES + common.B_CASES)
def test_oracle_sao_matches_reference(built, name):
    """TEncSampleAdaptiveOffset::SAOProcess: deblocking + SAO of the oracle on the pre-deblocking reconstruction must give the reference's
    finished picture, the same per-CTU SAO parameters and the same slice-level enable flags for every picture of the clip (the
    picture-level on/off rule carries the disabled rates from picture to picture)."""
    import oracle
    saod = {}
    cfg, slices, finals = common.load_ldp_case(name, sao=saod)
    rate = np.zeros((3, 8), np.float64)
    n_new = 0
    for r in slices:
        poc = int(r["poc"])
        ctus, ictus = common.split_fixture_ctus(r["ctus"])
        dbk = oracle.deblock(r["rec"], cfg["bit_depth"], int(r["qp"]), int(r["slice_type"]), r["ref_poc"], ctus, ictus)
        org = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], poc, cfg["seed"])
        a = saod[poc]
        out, params, en = oracle.sao(org, dbk, cfg["bit_depth"], int(r["qp"]), float(r["lambda"]), float(r["weight_cb"]), int(r["cabac_init_type"]), a["depth"], rate)
        assert (int(en[0]), int(en[1])) == tuple(a["enabled"]) and en[1] == en[2], f"{name} POC {poc}: slice-level SAO flags"
        assert np.array_equal(common.normalise_sao(params), common.normalise_sao(a["sao"])), f"{name} POC {poc}: SAO parameters"
        for c in range(3):
            assert np.array_equal(out[c], finals[poc]["rec"][c]), f"{name} POC {poc}: finished picture plane {c}"
        n_new += int((a["sao"][:, :, 0] == 1).sum())
    assert n_new > 0, "the fixture never chose new offsets"


def fixture_bits_args(cfg, r, sao_rec, with_sao):
    """keyword arguments of oracle.encode_slice / Encoder.encode_slices for one 'S' record of an inter fixture"""
    return dict(cabac_init_type=int(r["cabac_init_type"]), num_ref_idx=tuple(int(v) for v in r["num_ref_idx"]), mvd_l1_zero=int(r["mvd_l1_zero"]),
                max_merge_cand=int(r["max_merge_cand"]), sao=sao_rec["sao"] if with_sao else None, sao_enabled=tuple(sao_rec["enabled"]) if with_sao else (0, 0))


@pytest.mark.parametrize("name", common.LDP_CASES + common.B_CASES + common.DBK_CASES + common.LDP_LONG_CASES)
def test_oracle_bitstream_pass_matches_reference(built, name):
    """TEncSlice::encodeSlice: from the reference's own CTU decisions and SAO parameters the oracle's arithmetic coder must write the same
    substream bytes as the reference (I, P and B slices, 8 and 10 bit, one substream or one per CTU row, SAO syntax on and off), code the
    same number of bins and leave the same context table choice for the next picture (determineCabacInitIdx)."""
    import oracle
    sd, bd = {}, {}
    cfg, slices, _ = common.load_ldp_case(name, sao=sd, bits=bd)
    with_sao = not name.startswith("dbk_")
    total = 0
    for r in slices:
        poc, st = int(r["poc"]), int(r["slice_type"])
        ctus, ictus = common.split_fixture_ctus(r["ctus"])
        subs, nxt, bins = oracle.encode_slice(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], st, int(r["qp"]), ctus, ictus if st != 2 else None,
                                              **fixture_bits_args(cfg, r, sd.get(poc), with_sao))
        want = bd[poc]
        assert len(subs) == len(want["substreams"])
        for k, (g, w) in enumerate(zip(subs, want["substreams"])):
            assert g == w, f"{name} POC {poc}: substream {k} differs ({len(g)} vs {len(w)} bytes)"
        assert bins == want["num_bins"] and nxt == want["next_cabac_init_type"], f"{name} POC {poc}: bins {bins}/{want['num_bins']}, next table {nxt}/{want['next_cabac_init_type']}"
        total += sum(len(s) for s in subs)
    assert total > 500


@pytest.mark.parametrize("name", common.YUVIO_CASES)
def test_oracle_yuv_io_matches_reference(built, name):
    """TVideoIOYuv::read / ::write (picture ingest and output): the planes the reference's reader produced from raw file frames (8 -> 10 bit,
    10 -> 8 bit with rounding and clipping, padding by repetition) and the bytes its writer produced from them (conformance crop, down / up
    conversion) must equal the oracle's."""
    import oracle
    c = common.load_yuvio_case(name)
    for i in range(c["frames"]):
        planes = oracle.yuv_read(c["raw"][i], c["file_w"], c["file_h"], c["file_bd"], c["internal_bd"], c["pad_x"], c["pad_y"])
        for k in range(3):
            assert np.array_equal(planes[k], c["planes"][i][k]), f"{name} frame {i}: plane {k}"
        assert oracle.yuv_write(planes, c["internal_bd"], c["out_bd"], c["pad_x"], c["pad_y"]) == c["out"][i], f"{name} frame {i}: written bytes"


@pytest.mark.parametrize("name", [common.FULL_CASES[1], common.FULL_CASES[0]])
def test_oracle_matches_full_size_reference_digests(built, name):
    """the oracle at BASELINE.json's picture sizes against the reference's own full pictures (tests/gen_golden_full.py): every CTU of two
    1920x1080 10-bit I pictures (no WPP) and of one 3840x2160 10-bit WPP I picture, plus the reconstruction before the loop filters"""
    import oracle
    cfg, pics = common.load_full_case(name)
    for p in pics:
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(p["poc"]), cfg["seed"])
        rec, ctus = oracle.compress(planes, cfg["bit_depth"], int(p["qp"]), cfg["wpp"])
        got = np.zeros(len(ctus), __import__("hmd2").CTU_DT)
        for f in ("total_cost", "total_bits", "total_dist", "depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx", "cbf", "tskip",
                  "coeff_y", "coeff_cb", "coeff_cr"):
            got[f] = ctus[f]
        dig = common.ctu_digests(common.split_fixture_ctus(got)[0])
        bad = np.nonzero((dig != p["ctu_sha1"]).any(axis=1))[0]
        assert len(bad) == 0, f"{name} POC {int(p['poc'])}: {len(bad)} of {len(dig)} CTUs differ, first CTU {int(bad[0])}"
        for c in range(3):
            assert np.array_equal(common.md5_of(np.ascontiguousarray(rec[c], np.uint16)), p["rec_md5"][c]), f"{name} POC {int(p['poc'])}: reconstruction plane {c}"


@pytest.mark.parametrize("name", common.LCU_RC_CASES)
def test_oracle_lcu_level_rate_control_matches_reference(built, name):
    """SURVEY 8f n4, stage 2 (oracle first): clips the reference encoded with --RateControl=1 --LCULevelRateControl=1 (low-delay P with WPP, all-intra 10-bit,
    random access 10-bit).  TEncSlice::compressSlice asks the
    rate model for a lambda and a QP per CTU (TEncSlice.cpp:776-808: TComRdCost::setLambda, TComTrQuant::setLambdas, setRCQP); the harness records both
    ('L' record).  Given them the restated search reproduces decisions, motion, coefficients, costs, reconstruction, m_phQP and m_bEncodeDQP of every
    picture bit for bit -- the search side of the LCU-level rate control is pinned; the model itself (TEncRateCtrl.cpp) and a lambda per CTU on the
    device are what stage 2 still needs."""
    import oracle
    cfg, slices, finals = common.load_ldp_case(name)
    n_var = 0
    for r in slices:
        q, lcu = r["dqp"], r["lcu_rc"]
        assert q is not None and lcu is not None and int(q["max_cu_dqp_depth"]) == 0
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        rec, ctus, ictus, qp, flag = oracle.compress_dqp(planes, cfg["bit_depth"], r, finals, cfg["wpp"], lcu["ctu_qp"].astype(np.int8), int(q["dqp_flag_in"]),
                                                         ctu_lambda=lcu["ctu_lambda"])
        what = f"{name} POC {int(r['poc'])}"
        if ictus is None:
            common.assert_ctus_equal(ctus, common.split_fixture_ctus(r["ctus"])[0], what)
        else:
            common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], what)
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{what}: reconstruction plane {c}"
        m = common.inside_mask(len(ctus), cfg["width"], cfg["height"])
        assert np.array_equal(qp[m], q["qp"][m]), f"{what}: QP differs in CTUs {np.nonzero((qp != q['qp']).any(axis=1))[0][:8]}"
        assert flag == int(q["dqp_flag_out"]), f"{what}: m_bEncodeDQP after the slice"
        n_var += int(len(set(np.round(lcu["ctu_lambda"], 9))) > 1)
    assert n_var >= 2, "the clip should have pictures whose CTUs were searched with different lambdas"


@pytest.mark.parametrize("name", common.DQP_CASES)
def test_oracle_cu_qp_delta_matches_reference(built, name):
    """clips the reference encoded with AdaptiveQP (I, P and B slices, WPP on / off, 8 / 10 bit) and with the picture-level rate
    control.  The restated TEncPreanalyzer reproduces the reference's activities and xComputeQP its per-CTU QPs; with them the search
    reproduces decisions, motion, coefficients, costs, the reconstruction, TComDataCU::m_phQP and TEncCu::m_bEncodeDQP bit for bit."""
    import oracle
    cfg, slices, finals = common.load_ldp_case(name)
    n_q = 0
    for r in slices:
        q = r["dqp"]
        assert q is not None and int(q["max_cu_dqp_depth"]) == 0
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        ctu_qp = None
        if int(q["aq_range"]) > 0:
            act, avg = oracle.preanalyze(planes[0])
            assert np.array_equal(act, q["activity"]) and avg == float(q["avg_activity"]), f"{name} POC {int(r['poc'])}: activities"
            ctu_qp = oracle.aq_qp(act, avg, int(q["aq_range"]), int(r["qp"]), cfg["bit_depth"])
        rec, ctus, ictus, qp, flag = oracle.compress_dqp(planes, cfg["bit_depth"], r, finals, cfg["wpp"], ctu_qp, int(q["dqp_flag_in"]))
        what = f"{name} POC {int(r['poc'])}"
        if ictus is None:
            common.assert_ctus_equal(ctus, common.split_fixture_ctus(r["ctus"])[0], what)
        else:
            common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], what)
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{what}: reconstruction plane {c}"
        m = common.inside_mask(len(ctus), cfg["width"], cfg["height"])
        assert np.array_equal(qp[m], q["qp"][m]), f"{what}: QP differs in CTUs {np.nonzero((qp != q['qp']).any(axis=1))[0][:8]}"
        assert flag == int(q["dqp_flag_out"]), f"{what}: m_bEncodeDQP after the slice"
        n_q += 1
    assert n_q >= 2
