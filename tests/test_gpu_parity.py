"""GPU suite (-m gpu): the HIP path through the C ABI against (1) the golden vectors of the real
reference encoder, (2) the oracle on freshly seeded inputs, (3) the primitive KATs."""
import numpy as np
import pytest

import common
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hm():
    import hm355
    return hm355


@pytest.mark.parametrize("name", common.CASES)
def test_hip_matches_reference_fixture(hm, name):
    cfg, frames = common.load_case(name)
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=cfg["frames"])
    planes = [synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], i, cfg["seed"]) for i in range(cfg["frames"])]
    res = enc.compress(planes, cfg["qp"])
    for i, (ctus, rec) in enumerate(frames):
        got_rec, got_ctus, stats = res[i]
        common.assert_ctus_equal(got_ctus, ctus, f"{name} frame {i}", (cfg["width"], cfg["height"]))
        common.assert_rec_equal(got_rec, rec, cfg["width"], cfg["height"], f"{name} frame {i}")
        assert stats[0] == int(ctus["total_bits"].sum()) and stats[2] == int(ctus["total_dist"].sum())
    enc.close()


@pytest.mark.parametrize("w,h,bd,qp,wpp,seed", [(192, 128, 8, 30, 1, 11), (128, 72, 10, 34, 0, 12), (64, 64, 8, 25, 0, 13),
                                                   # dense and nearly empty 32x32 blocks: RDOQ state of 32x32 blocks, early terminations of the CU / residual quadtrees
                                                   (448, 256, 10, 12, 1, 21), (320, 192, 8, 47, 0, 22), (384, 192, 10, 22, 1, 23), (200, 192, 8, 3, 0, 24)])
def test_hip_matches_oracle_on_fresh_inputs(built, hm, w, h, bd, qp, wpp, seed):
    import oracle
    planes = synth.frame(w, h, bd, 0, seed)
    want_rec, want_ctus = oracle.compress(planes, bd, qp, wpp)
    enc = hm.Encoder(w, h, bd, wpp, max_batch=1)
    (got_rec, got_ctus, _), = enc.compress([planes], qp)
    common.assert_ctus_equal(got_ctus, want_ctus, f"{w}x{h}")
    for k in range(3):
        assert np.array_equal(got_rec[k], want_rec[k])
    # bitstream pass on the slot the search just filled, against the oracle's arithmetic coder on the same CTU data
    (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=2, qp=qp)])
    want_subs, want_nxt, want_bins = oracle.encode_slice(w, h, bd, wpp, 2, qp, want_ctus)
    assert subs == want_subs and nxt == want_nxt and bins == want_bins
    assert len(subs) == ((h + 63) // 64 if wpp else 1) and all(len(x) > 0 for x in subs)
    enc.close()


@pytest.mark.parametrize("team", ["0", "1"])
@pytest.mark.parametrize("name", common.LDP_CASES + common.B_CASES)
def test_hip_p_and_b_slices_match_reference_fixture(hm, monkeypatch, name, team):
    """low-delay P, random access and low-delay B clips: every P / B slice through hm355_compress_slice_inter with the reference
    pictures and slice parameters the reference's compressSlice saw; decisions, motion, coefficients, costs, reconstruction bit-exact --
    searched by one wavefront per CTU (team 0) and by a team of wavefronts per CTU (team 1, hm355_team.h: what a one-picture launch gets)."""
    monkeypatch.setenv("HM355_TEAM", team)
    cfg, slices, finals = common.load_ldp_case(name)
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=1)
    n_p = 0
    for r in slices:
        if int(r["slice_type"]) == 2:
            continue
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        sp, refs = common.ldp_slice_inputs(r, finals)
        rec, ctus, ictus, stats = enc.compress_inter(planes, sp, refs)
        common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], f"{name} POC {int(r['poc'])}")
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{name} POC {int(r['poc'])}: reconstruction plane {c}"
        assert stats[0] == int(ctus["total_bits"].sum())
        n_p += 1
    assert n_p >= 3
    enc.close()


def test_hip_p_slice_batch_equals_single(hm):
    """P pictures of one call are independent: all P slices of a clip in one batch == the fixture (each with its own references)"""
    name = common.LDP_CASES[2]
    cfg, slices, finals = common.load_ldp_case(name)
    ps = [r for r in slices if int(r["slice_type"]) == 1]
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=len(ps))
    jobs = []
    for r in ps:
        sp, refs = common.ldp_slice_inputs(r, finals)
        jobs.append((synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"]), sp, refs))
    for r, (rec, ctus, ictus, stats) in zip(ps, enc.compress_inter_batch(jobs)):
        common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], f"{name} POC {int(r['poc'])} (batched)")
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c])
    enc.close()


@pytest.mark.parametrize("w,h,bd,wpp,kind", [(416, 240, 8, 1, "P"), (264, 200, 10, 0, "B"), (416, 240, 10, 1, "B")])
def test_hip_inter_matches_oracle_on_fresh_inputs(built, hm, w, h, bd, wpp, kind):
    """P / B slices on freshly seeded pictures (no fixture): reference pictures = the HIP path's own I-slice reconstructions, two pictures
    per list; the HIP result must equal the oracle's bit for bit (decisions, motion, coefficients, costs, reconstruction)."""
    import math
    import oracle
    qp, seed = 30, 77
    enc = hm.Encoder(w, h, bd, wpp, max_batch=2)
    n = enc.num_ctus
    res = enc.compress([synth.frame(w, h, bd, f, seed) for f in (0, 4)], qp)
    mot = np.zeros(n, [("pred_mode", "u1", 256), ("mv0", "<i2", (256, 2)), ("ref_idx0", "i1", 256), ("mv1", "<i2", (256, 2)), ("ref_idx1", "i1", 256)])
    mot["pred_mode"] = 1; mot["ref_idx0"] = -1; mot["ref_idx1"] = -1
    zero = np.zeros((2, 16), np.int32)
    finals = {poc: {"poc": poc, "slice_type": 2, "rec": res[k][0], "motion": mot, "num_ref_idx": (0, 0), "ref_poc": zero, "ref_long_term": zero}
              for k, poc in enumerate((0, 4))}
    ref_poc = np.zeros((2, 16), np.int32)
    ref_poc[0, :2] = (0, 4)
    if kind == "B":
        ref_poc[1, :2] = (4, 0)
    lam = 0.4624 * 2.0 ** ((qp + 2 - 12) / 3.0) * 2.0
    srec = {"poc": 2, "slice_type": 1 if kind == "P" else 0, "qp": qp + 2, "lambda": lam, "weight_cb": hm.intra_lambda(qp + 2)[1],
            "cabac_init_type": 1 if kind == "P" else 0, "num_ref_idx": (2, 0 if kind == "P" else 2), "ref_poc": ref_poc, "col_from_l0": 0 if kind == "B" else 1,
            "col_ref_idx": 0, "tmvp": 1, "mvd_l1_zero": 0, "max_merge_cand": 5, "check_ldc": 0 if kind == "B" else 1,
            "lambda_motion_sad": int(math.floor(65536.0 * math.sqrt(lam))), "lambda_motion_sse": int(math.floor(65536.0 * lam))}
    cur = synth.frame(w, h, bd, 2, seed)
    want_rec, want_ctus, want_ictus = oracle.compress_inter(cur, bd, srec, finals, wpp=wpp)
    sp, refs = common.ldp_slice_inputs(srec, finals)
    rec, ctus, ictus, _ = enc.compress_inter(cur, sp, refs)
    enc.close()
    for f in ("total_bits", "total_dist", "total_cost", "depth", "part_size", "pred_mode", "tr_idx", "cbf", "tskip", "coeff_y", "coeff_cb", "coeff_cr"):
        assert np.array_equal(ctus[f], want_ctus[f]), f"{kind} {w}x{h}: {f} differs"
    for f in ("skip", "merge_flag", "merge_idx", "inter_dir", "mv", "mvd", "ref_idx", "mvp_idx", "mvp_num"):
        assert np.array_equal(ictus[f], want_ictus[f]), f"{kind} {w}x{h}: {f} differs"
    for c in range(3):
        assert np.array_equal(rec[c], want_rec[c])
    assert (ictus["inter_dir"] != 0).any()


def test_hip_slot_reuse_across_slice_types(hm):
    """a slot used for an inter slice carries nothing over into a later I slice (and back), and P and B slices can share one batch"""
    cfgp, slp, finp = common.load_ldp_case(common.LDP_CASES[0])
    w, h, bd = cfgp["width"], cfgp["height"], cfgp["bit_depth"]
    enc = hm.Encoder(w, h, bd, 0, max_batch=2)
    i_planes = synth.frame(w, h, bd, 0, cfgp["seed"])
    (rec_a, ctus_a, _), = enc.compress([i_planes], 32)
    rp = [r for r in slp if int(r["slice_type"]) == 1][1]
    sp, refs = common.ldp_slice_inputs(rp, finp)
    p_planes = synth.frame(w, h, bd, int(rp["poc"]), cfgp["seed"])
    # the same P slice twice in one batch, once declared as a B slice whose list 1 repeats list 0's first picture (different search, same entry point)
    spb = dict(sp); spb["slice_type"] = 0; spb["cabac_init_type"] = 0
    spb["num_ref_idx"] = (sp["num_ref_idx"][0], 1)
    spb["ref_poc"] = np.array(sp["ref_poc"]); spb["ref_poc"][1][0] = spb["ref_poc"][0][0]
    out = enc.compress_inter_batch([(p_planes, sp, refs), (p_planes, spb, refs)])
    common.assert_inter_ctus_equal(out[0][1], out[0][2], rp["ctus"], "P slice next to a B slice in one batch")
    assert (out[1][2]["inter_dir"] == 3).any(), "the B variant found no bi-predicted partition"
    (rec_b, ctus_b, _), = enc.compress([i_planes], 32)
    common.assert_ctus_equal(ctus_b, ctus_a, "I slice after inter slices in the same slot")
    for c in range(3):
        assert np.array_equal(rec_a[c], rec_b[c])
    rec, ctus, ictus, _ = enc.compress_inter(p_planes, sp, refs)
    common.assert_inter_ctus_equal(ctus, ictus, rp["ctus"], "P slice after an I slice in the same slot")
    enc.close()


def test_hip_inter_slice_rejects_bad_parameters(hm):
    cfg, slices, finals = common.load_ldp_case(common.LDP_CASES[0])
    r = [s for s in slices if int(s["slice_type"]) == 1][0]
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], 0, max_batch=1)
    sp, refs = common.ldp_slice_inputs(r, finals)
    sp["num_ref_idx"] = (sp["num_ref_idx"][0], 1)                  # a P slice has no list 1
    sp["ref_poc"] = np.array(sp["ref_poc"]); sp["ref_poc"][1][0] = sp["ref_poc"][0][0]
    with pytest.raises(RuntimeError):
        enc.compress_inter(synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"]), sp, refs)
    enc.close()


@pytest.mark.parametrize("name", common.DBK_CASES)
def test_hip_deblocking_matches_reference(hm, name):
    """hm355_deblock on the reference's pre-deblocking reconstruction + per-CTU data == the reference's deblocked picture (SAO off),
    for I, P and B slices; and the device-resident path (compress, then hm355_deblock_run in place, then download) gives the same."""
    cfg, slices, finals = common.load_ldp_case(name)
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], 0, max_batch=1)
    for r in slices:
        ctus, ictus = common.split_fixture_ctus(r["ctus"])
        st = int(r["slice_type"])
        got = enc.deblock(r["rec"], st, int(r["qp"]), r["ref_poc"], ctus, ictus if st != 2 else None)
        want = finals[int(r["poc"])]["rec"]
        for c in range(3):
            assert np.array_equal(got[c], want[c]), f"{name} POC {int(r['poc'])}: deblocked plane {c} differs at {int((got[c] != want[c]).sum())} samples"
        # device-resident: search on the device, deblock in place, download
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], int(r["poc"]), cfg["seed"])
        if st == 2:
            enc.upload(0, planes)
            lam, cw = float(r["lambda"]), float(r["weight_cb"])
            sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, int(r["qp"]), lam, cw))
            enc._check(enc.lib.hm355_run(enc.h_, 1, sl), "hm355_run")
        else:
            sp, refs = common.ldp_slice_inputs(r, finals)
            enc.compress_inter(planes, sp, refs)
        enc.deblock_run([(st, int(r["qp"]), r["ref_poc"])])
        rec2, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(rec2[c], want[c]), f"{name} POC {int(r['poc'])}: device-resident deblocking differs in plane {c}"
    enc.close()


@pytest.mark.parametrize("name", common.DBK_CASES + [common.LDP_CASES[0], common.B_CASES[0]])
def test_hip_closed_loop_on_device_matches_reference(hm, name):
    """A whole clip on the device, no host round trip of pictures between frames: search -> deblock in place -> device-resident
    reference (border extension + compressMotion on the device) -> next picture's search.  Clips the reference ran with SAO off compare
    the deblocked picture; clips with the default loop filters also run hm355_sao_run and compare the finished picture, so the whole
    per-picture pipeline of the reference's default configuration runs on the device."""
    saod, bitd = {}, {}
    cfg, slices, finals = common.load_ldp_case(name, sao=saod, bits=bitd)
    with_sao = not name.startswith("dbk_")
    rate = np.zeros((3, 8), np.float64)
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=1)
    dev_refs = {}
    for r in slices:
        st, poc = int(r["slice_type"]), int(r["poc"])
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], poc, cfg["seed"])
        if st == 2:
            enc.upload(0, planes)
            sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, int(r["qp"]), float(r["lambda"]), float(r["weight_cb"])))
            enc._check(enc.lib.hm355_run(enc.h_, 1, sl), "hm355_run")
            rec, ctus, _ = enc.download(0)
            common.assert_ctus_equal(ctus, common.split_fixture_ctus(r["ctus"])[0], f"{name} POC {poc}")
        else:
            sp, _ = common.ldp_slice_inputs(r, finals)
            refs = {int(p): dev_refs[int(p)] for l in range(2) for p in r["ref_poc"][l][:r["num_ref_idx"][l]]}
            rec, ctus, ictus, _ = enc.compress_inter(planes, sp, refs)
            common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], f"{name} POC {poc} (device-resident references)")
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{name} POC {poc}: pre-deblocking reconstruction plane {c}"
        enc.deblock_run([(st, int(r["qp"]), r["ref_poc"])])
        en = (0, 0)
        if with_sao:
            (en3, _), = enc.sao_run([dict(qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), depth=saod[poc]["depth"], disabled_rate=rate,
                                          chroma_weight=float(r["weight_cb"]), **{"lambda": float(r["lambda"])})])
            en = (en3[0], en3[1])
        # the slice data of the picture from what the search and SAO left on the device (TEncGOP.cpp:1559 runs encodeSlice at this point)
        (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=st, qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), num_ref_idx=r["num_ref_idx"],
                                                         mvd_l1_zero=int(r["mvd_l1_zero"]), max_merge_cand=int(r["max_merge_cand"]), sao_enabled=en)])
        assert subs == bitd[poc]["substreams"], f"{name} POC {poc}: slice data bytes differ"
        assert (nxt, bins) == (bitd[poc]["next_cabac_init_type"], bitd[poc]["num_bins"]), f"{name} POC {poc}: next context table / bin count"
        dbk, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(dbk[c], finals[poc]["rec"][c]), f"{name} POC {poc}: finished picture plane {c}"
        dev_refs[poc] = enc.ref_from_slot(0, poc, st != 2, r["num_ref_idx"], r["ref_poc"], r["ref_long_term"])
    for ref in dev_refs.values():
        enc.ref_release(ref)
    enc.close()


@pytest.mark.parametrize("name", common.DQP_CASES)
def test_hip_cu_qp_delta_matches_reference(hm, name):
    """the clips the reference encoded with AdaptiveQP (I / P / B, WPP on and off) and with the picture-level rate control, closed loop
    on the device: hm355_preanalyze + the reference's double arithmetic give the reference's activities and per-CTU QPs; the search with
    hm355_set_dqp reproduces decisions, motion, coefficients, costs, reconstruction, TComDataCU::m_phQP and TEncCu::m_bEncodeDQP; deblocking with
    the CUs' QPs, SAO and the bitstream pass with the cu_qp_delta syntax reproduce the finished picture and the slice data bytes."""
    saod, bitd = {}, {}
    cfg, slices, finals = common.load_ldp_case(name, sao=saod, bits=bitd)
    w, h, bd = cfg["width"], cfg["height"], cfg["bit_depth"]
    rate = np.zeros((3, 8), np.float64)
    enc = hm.Encoder(w, h, bd, cfg["wpp"], max_batch=1)
    dev_refs = {}
    for r in slices:
        st, poc, q = int(r["slice_type"]), int(r["poc"]), r["dqp"]
        what = f"{name} POC {poc}"
        planes = synth.frame(w, h, bd, poc, cfg["seed"])
        enc.upload(0, planes)
        ctu_qp = None
        if int(q["aq_range"]) > 0:
            act, avg = hm.aq_activities(enc.preanalyze(0))
            assert np.array_equal(act, q["activity"]) and avg == float(q["avg_activity"]), f"{what}: activities"
            ctu_qp = hm.aq_ctu_qp(act, avg, int(q["aq_range"]), int(r["qp"]), bd)
        enc.set_dqp(0, ctu_qp, int(q["dqp_flag_in"]))
        if st == 2:
            sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, int(r["qp"]), float(r["lambda"]), float(r["weight_cb"])))
            enc._check(enc.lib.hm355_run(enc.h_, 1, sl), "hm355_run")
            rec, ctus, _ = enc.download(0)
            common.assert_ctus_equal(ctus, common.split_fixture_ctus(r["ctus"])[0], what)
        else:
            sp, _ = common.ldp_slice_inputs(r, finals)
            refs = {int(p): dev_refs[int(p)] for l in range(2) for p in r["ref_poc"][l][:r["num_ref_idx"][l]]}
            rec, ctus, ictus, _ = enc.compress_inter(planes, sp, refs)
            common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], what)
        for c in range(3):
            assert np.array_equal(rec[c], r["rec"][c]), f"{what}: pre-deblocking reconstruction plane {c}"
        qp, flag = enc.get_dqp(0)
        m = common.inside_mask(len(ctus), w, h)
        assert np.array_equal(qp[m], q["qp"][m]), f"{what}: m_phQP differs in CTUs {np.nonzero(((qp != q['qp']) & m).any(axis=1))[0][:8]}"
        assert flag == int(q["dqp_flag_out"]), f"{what}: m_bEncodeDQP after the slice"
        enc.deblock_run([(st, int(r["qp"]), r["ref_poc"])])
        (en3, _), = enc.sao_run([dict(qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), depth=saod[poc]["depth"], disabled_rate=rate,
                                      chroma_weight=float(r["weight_cb"]), **{"lambda": float(r["lambda"])})])
        (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=st, qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), num_ref_idx=r["num_ref_idx"],
                                                         mvd_l1_zero=int(r["mvd_l1_zero"]), max_merge_cand=int(r["max_merge_cand"]), sao_enabled=(en3[0], en3[1]))])
        assert subs == bitd[poc]["substreams"], f"{what}: slice data bytes differ"
        assert (nxt, bins) == (bitd[poc]["next_cabac_init_type"], bitd[poc]["num_bins"]), f"{what}: next context table / bin count"
        fin, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(fin[c], finals[poc]["rec"][c]), f"{what}: finished picture plane {c}"
        dev_refs[poc] = enc.ref_from_slot(0, poc, st != 2, r["num_ref_idx"], r["ref_poc"], r["ref_long_term"])
    for ref in dev_refs.values():
        enc.ref_release(ref)
    enc.close()


def _first_bad(got, want):
    bad = np.nonzero((got != want).any(axis=1))[0]
    return f"{len(bad)} CTUs differ, first CTU {int(bad[0])}" if len(bad) else ""


@pytest.mark.parametrize("name", common.FULL_CASES)
def test_full_size_pictures_match_reference_digests(hm, name):
    """BASELINE.json's sizes against the real reference, every CTU of every picture: the reference encoded these clips in the development
    container (tests/gen_golden_full.py) and the fixture holds, per picture, a SHA-1 per CTU over decisions / costs / coefficients / motion,
    MD5s of the reconstruction before the loop filters and of the finished picture, of the SAO parameters and of every substream of the slice
    data.  The whole per-picture pipeline runs on the device in coding order (search -> deblock -> SAO -> bitstream pass -> device-resident
    reference for the following pictures): 3840x2160 10-bit WPP I picture; two 1920x1080 10-bit I pictures without WPP (the CABAC state
    chains through all 510 CTUs); 1080p low-delay I + 2 P; 4K random access POC 0 / 8 / 4 (B slices, references in both directions)."""
    cfg, pics = common.load_full_case(name)
    w, h, bd = cfg["width"], cfg["height"], cfg["bit_depth"]
    enc = hm.Encoder(w, h, bd, cfg["wpp"], max_batch=1)
    rate = np.zeros((3, 8), np.float64)
    dev_refs = {}
    for p in pics:
        st, poc, qp = int(p["slice_type"]), int(p["poc"]), int(p["qp"])
        what = f"{name} POC {poc}"
        planes = synth.frame(w, h, bd, poc, cfg["seed"])
        if st == 2:
            enc.upload(0, planes)
            sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, qp, float(p["lambda"]), float(p["weight_cb"])))
            enc._check(enc.lib.hm355_run(enc.h_, 1, sl), "hm355_run")
            rec, ctus, _ = enc.download(0)
            dig = common.ctu_digests(ctus)
        else:
            sp = {k: p[k] for k in ("slice_type", "qp", "lambda", "poc", "cabac_init_type", "num_ref_idx", "ref_poc", "col_from_l0", "col_ref_idx", "tmvp",
                                    "mvd_l1_zero", "max_merge_cand", "check_ldc", "lambda_motion_sad", "lambda_motion_sse")}
            sp["chroma_weight"] = p["weight_cb"]
            refs = {int(q): dev_refs[int(q)] for l in range(2) for q in p["ref_poc"][l][:p["num_ref_idx"][l]]}
            rec, ctus, ictus, _ = enc.compress_inter(planes, sp, refs)
            dig = common.ctu_digests(ctus, ictus)
        assert np.array_equal(dig, p["ctu_sha1"]), f"{what}: {_first_bad(dig, p['ctu_sha1'])} of {len(dig)}"
        for c in range(3):
            assert np.array_equal(common.md5_of(rec[c]), p["rec_md5"][c]), f"{what}: pre-deblocking reconstruction plane {c}"
        enc.deblock_run([(st, qp, p["ref_poc"])])
        (en3, sao), = enc.sao_run([dict(qp=qp, cabac_init_type=int(p["cabac_init_type"]), depth=int(p["sao_depth"]), disabled_rate=rate,
                                        chroma_weight=float(p["weight_cb"]), **{"lambda": float(p["lambda"])})])
        assert (en3[0], en3[1]) == tuple(int(v) for v in p["sao_enabled"]), f"{what}: slice-level SAO flags"
        assert np.array_equal(common.md5_of(common.normalise_sao(sao)), p["sao_md5"]), f"{what}: SAO parameters"
        (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=st, qp=qp, cabac_init_type=int(p["cabac_init_type"]), num_ref_idx=p["num_ref_idx"],
                                                         mvd_l1_zero=int(p["mvd_l1_zero"]), max_merge_cand=int(p["max_merge_cand"]), sao_enabled=(en3[0], en3[1]))])
        assert [len(x) for x in subs] == [int(v) for v in p["sub_sizes"]], f"{what}: substream sizes"
        for k, x in enumerate(subs):
            assert np.array_equal(common.md5_of(np.frombuffer(x, np.uint8)), p["sub_md5"][k]), f"{what}: substream {k}"
        assert (nxt, bins) == (int(p["next_cabac_init_type"]), int(p["num_bins"])), f"{what}: next context table / bin count"
        fin, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(common.md5_of(fin[c]), p["final_md5"][c]), f"{what}: finished picture plane {c}"
        dev_refs[poc] = enc.ref_from_slot(0, poc, st != 2, p["num_ref_idx"], p["ref_poc"], p["ref_long_term"])
    for ref in dev_refs.values():
        enc.ref_release(ref)
    enc.close()


@pytest.mark.parametrize("name", common.LDP_CASES + common.B_CASES + common.DBK_CASES + common.LDP_LONG_CASES)
def test_hip_bitstream_pass_matches_reference(hm, name):
    """hm355_encode_slice (host buffers in) on the reference's own CTU decisions and SAO parameters: the substream bytes, the bin count and
    the context table choice for the next picture must equal what the reference's TEncSlice::encodeSlice produced (I, P and B slices,
    8 and 10 bit, one substream or one per CTU row, with and without SAO syntax)."""
    sd, bd = {}, {}
    cfg, slices, _ = common.load_ldp_case(name, sao=sd, bits=bd)
    with_sao = not name.startswith("dbk_")
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=1)
    for r in slices:
        poc, st = int(r["poc"]), int(r["slice_type"])
        ctus, ictus = common.split_fixture_ctus(r["ctus"])
        a = sd.get(poc)
        desc = dict(slice_type=st, qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), num_ref_idx=r["num_ref_idx"], mvd_l1_zero=int(r["mvd_l1_zero"]),
                    max_merge_cand=int(r["max_merge_cand"]), sao_enabled=tuple(a["enabled"]) if with_sao else (0, 0))
        subs, nxt, bins = enc.encode_slice(desc, ctus, ictus if st != 2 else None, a["sao"] if with_sao else None)
        want = bd[poc]
        for k, (g, w) in enumerate(zip(subs, want["substreams"])):
            assert g == w, f"{name} POC {poc}: substream {k} differs ({len(g)} vs {len(w)} bytes)"
        assert len(subs) == len(want["substreams"]) and (nxt, bins) == (want["next_cabac_init_type"], want["num_bins"]), f"{name} POC {poc}"
    enc.close()


@pytest.mark.parametrize("name", common.YUVIO_CASES)
def test_hip_picture_ingest_and_output_match_reference(hm, name):
    """hm355_upload_file_frames / hm355_download_file_frames (TVideoIOYuv::read / ::write on the device) against the reference's own reader and
    writer: raw file frames -> the slot's original planes (bit-depth scaling, padding by repetition), and those planes -> file bytes
    (conformance crop, rounding and clipping); both frames of a fixture go through one batched launch."""
    c = common.load_yuvio_case(name)
    enc = hm.Encoder(c["width"], c["height"], c["internal_bd"], 0, max_batch=c["frames"])
    enc.upload_file_frames(c["raw"], c["file_w"], c["file_h"], c["file_bd"])
    for i in range(c["frames"]):
        got = enc.download_org(i)
        for k in range(3):
            assert np.array_equal(got[k], c["planes"][i][k]), f"{name} frame {i}: plane {k} differs at {int((got[k] != c['planes'][i][k]).sum())} samples"
    out, _ = enc.download_file_frames(c["frames"], c["out_bd"], c["pad_x"], c["pad_y"], source=1)
    for i in range(c["frames"]):
        assert out[i] == c["out"][i], f"{name} frame {i}: written bytes differ"
    enc.close()


def test_hip_ingest_search_output_chain_matches_oracle(built, hm):
    """file frame -> ingest -> search -> reconstruction as a file frame, all on the device, against the oracle chain (8-bit file, 10-bit
    internal, picture padded from 180x100 to 184x104, reconstruction written back at 8 bit through the conformance window)"""
    import oracle
    fw, fh, px, py, qp = 180, 100, 4, 4, 30
    base = synth.frame(fw + px, fh + py, 8, 0, 77)
    raw = b"".join(p[:fh >> (1 if k else 0), :fw >> (1 if k else 0)].astype(np.uint8).tobytes() for k, p in enumerate(base))
    planes = oracle.yuv_read(raw, fw, fh, 8, 10, px, py)
    want_rec, want_ctus = oracle.compress(planes, 10, qp, 0)
    enc = hm.Encoder(fw + px, fh + py, 10, 0, max_batch=1)
    enc.upload_file_frames([raw], fw, fh, 8)
    enc.run(1, qp)
    rec, ctus, _ = enc.download(0)
    common.assert_ctus_equal(ctus, want_ctus, "ingested picture")
    (out,), _ = enc.download_file_frames(1, 8, px, py)
    assert out == oracle.yuv_write(want_rec, 10, 8, px, py)
    with pytest.raises(RuntimeError):
        enc.upload_file_frames([raw + raw], fw * 2, fh, 8)            # wider than the configured picture
    enc.close()


@pytest.mark.parametrize("wpp", [0, 1])
def test_hip_bitstream_pass_batch_matches_oracle(built, hm, wpp):
    """six different pictures resident in six slots, one bitstream-pass launch (items = rows x pictures, row-major): every picture's
    substreams equal the oracle's on the same search results, and equal the picture coded alone"""
    import oracle
    w, h, bd, qp, n = 192, 136, 8, 28, 6
    enc = hm.Encoder(w, h, bd, wpp, max_batch=n)
    pics = [synth.frame(w, h, bd, i, 300 + i) for i in range(n)]
    res = enc.compress(pics, qp)
    got = enc.encode_slices_run([dict(slice_type=2, qp=qp)] * n)
    for i in range(n):
        oc = np.zeros(len(res[i][1]), oracle.CTU_DTYPE)
        for f in oc.dtype.names:
            oc[f] = res[i][1][f]
        want = oracle.encode_slice(w, h, bd, wpp, 2, qp, oc)
        assert got[i][0] == want[0] and got[i][1:] == want[1:], f"picture {i} of the batch"
    enc.close()
    one = hm.Encoder(w, h, bd, wpp, max_batch=1)
    one.compress([pics[3]], qp)
    (alone,) = one.encode_slices_run([dict(slice_type=2, qp=qp)])
    assert alone == got[3]
    one.close()


def test_hip_bitstream_pass_more_substreams_than_resident_workgroups(built, hm):
    """2,720 substreams (80 pictures of 2 x 34 CTUs, WPP) in one launch: more than the workgroups the GPU keeps resident (about 2,300), with
    the WPP hand-off between the rows of every picture -- the ticket order of the kernel must drain them; every picture against the oracle"""
    import oracle
    w, h, bd, qp, n = 128, 2176, 8, 36, 80
    enc = hm.Encoder(w, h, bd, 1, max_batch=n)
    pics = [synth.frame(w, h, bd, i % 5, 900 + i % 5) for i in range(n)]
    res = enc.compress(pics, qp)
    got = enc.encode_slices_run([dict(slice_type=2, qp=qp)] * n)
    enc.close()
    want = {}
    for i in range(n):
        if i % 5 not in want:
            oc = np.zeros(len(res[i][1]), oracle.CTU_DTYPE)
            for f in oc.dtype.names:
                oc[f] = res[i][1][f]
            want[i % 5] = oracle.encode_slice(w, h, bd, 1, 2, qp, oc)
        assert len(got[i][0]) == 34 and got[i][0] == want[i % 5][0] and got[i][1:] == want[i % 5][1:], f"picture {i}"


def test_hip_bitstream_pass_rejects_bad_parameters(hm):
    enc = hm.Encoder(128, 64, 8, 0, max_batch=1)
    with pytest.raises(RuntimeError):
        enc.encode_slices_run([dict(slice_type=1, qp=30, cabac_init_type=1, num_ref_idx=(1, 0))])       # no motion data in the slot
    with pytest.raises(RuntimeError):
        enc.encode_slices_run([dict(slice_type=2, qp=30, sao_enabled=(1, 1))])                           # no SAO parameters in the slot
    with pytest.raises(RuntimeError):
        enc.encode_slices_run([dict(slice_type=2, qp=77)])
    enc.close()


@pytest.mark.parametrize("name", [common.LDP_CASES[1], common.LDP_CASES[2], common.B_CASES[0], common.B_CASES[1]])
def test_hip_deblock_and_sao_match_reference(hm, name):
    """search -> hm355_deblock_run -> hm355_sao_run on the device, picture by picture of a clip the reference encoded with its default loop
    filters: the finished picture, the per-CTU SAO parameters and the slice-level SAO flags must equal the reference's (the disabled
    rates are carried from picture to picture as TEncSampleAdaptiveOffset does)."""
    saod = {}
    cfg, slices, finals = common.load_ldp_case(name, sao=saod)
    enc = hm.Encoder(cfg["width"], cfg["height"], cfg["bit_depth"], cfg["wpp"], max_batch=1)
    rate = np.zeros((3, 8), np.float64)
    for r in slices:
        st, poc = int(r["slice_type"]), int(r["poc"])
        planes = synth.frame(cfg["width"], cfg["height"], cfg["bit_depth"], poc, cfg["seed"])
        if st == 2:
            enc.upload(0, planes)
            sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, int(r["qp"]), float(r["lambda"]), float(r["weight_cb"])))
            enc._check(enc.lib.hm355_run(enc.h_, 1, sl), "hm355_run")
        else:
            sp, refs = common.ldp_slice_inputs(r, finals)
            enc.compress_inter(planes, sp, refs)
        enc.deblock_run([(st, int(r["qp"]), r["ref_poc"])])
        a = saod[poc]
        (en, params), = enc.sao_run([dict(qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), depth=a["depth"], disabled_rate=rate,
                                          chroma_weight=float(r["weight_cb"]), **{"lambda": float(r["lambda"])})])
        assert (en[0], en[1]) == tuple(a["enabled"]) and en[1] == en[2], f"{name} POC {poc}: slice-level SAO flags {en} vs {a['enabled']}"
        assert np.array_equal(common.normalise_sao(params), common.normalise_sao(a["sao"])), f"{name} POC {poc}: SAO parameters"
        out, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(out[c], finals[poc]["rec"][c]), f"{name} POC {poc}: finished picture plane {c} differs at {int((out[c] != finals[poc]['rec'][c]).sum())} samples"
    enc.close()


def test_hip_batch_equals_single(hm):
    """pictures of a batch are independent: batched results == one-at-a-time results"""
    w, h, bd, qp = 128, 128, 8, 32
    planes = [synth.frame(w, h, bd, i, 5) for i in range(3)]
    enc = hm.Encoder(w, h, bd, 1, max_batch=3)
    batch = enc.compress(planes, qp)
    for i in range(3):
        (rec, ctus, _), = enc.compress([planes[i]], qp)
        common.assert_ctus_equal(batch[i][1], ctus, f"picture {i}")
        for k in range(3):
            assert np.array_equal(batch[i][0][k], rec[k])
    enc.close()


@pytest.mark.parametrize("w,h,bd,qp,wpp,seed", [(256, 192, 10, 32, 1, 41), (200, 136, 8, 26, 0, 42), (448, 256, 8, 38, 1, 43)])
def test_hip_team_search_equals_single_wavefront_search(hm, monkeypatch, w, h, bd, qp, wpp, seed):
    """hm355_team.h: a CTU searched by a team of wavefronts (the unsplit candidate of every CU depth and the 2Nx2N candidate of the 8x8 CUs
    on helper wavefronts) gives what one wavefront gives -- decisions, coefficients, costs, reconstruction, CABAC hand-off (several
    pictures, partial CTUs at the right / bottom edge, with and without WaveFrontSynchro)."""
    planes = [synth.frame(w, h, bd, i, seed) for i in range(3)]
    enc = hm.Encoder(w, h, bd, wpp, max_batch=3)
    monkeypatch.setenv("HM355_TEAM", "0")
    one = enc.compress(planes, qp)
    monkeypatch.setenv("HM355_TEAM", "1")
    team = enc.compress(planes, qp)
    for i in range(3):
        common.assert_ctus_equal(team[i][1], one[i][1], f"picture {i}")
        for k in range(3):
            assert np.array_equal(team[i][0][k], one[i][0][k])
        assert team[i][2] == one[i][2]
    enc.close()


def test_hip_large_batch_code_path_equals_small_batch(hm):
    """A launch that can fill the device takes the fewest-instructions code path (Params::fewWaves == 0), a small one the shortest
    dependency chain; both must give the same result: 96 pictures in one batch == the same pictures in batches of 3."""
    w, h, bd, qp = 128, 64, 10, 27
    planes = [synth.frame(w, h, bd, i % 6, 21) for i in range(96)]
    enc = hm.Encoder(w, h, bd, 1, max_batch=96)
    big = enc.compress(planes, qp)
    small = enc.compress(planes[:3], qp) + enc.compress(planes[3:6], qp)
    for i in range(96):
        common.assert_ctus_equal(big[i][1], small[i % 6][1], f"picture {i}")
        for k in range(3):
            assert np.array_equal(big[i][0][k], small[i % 6][0][k])
    enc.close()


def test_full_size_4k_wpp_properties(built, hm):
    """BASELINE.json's full size (3840x2160 10-bit, WaveFrontSynchro=1), checked through properties that do not need
    the oracle at that size: (1) the same picture in two batch slots gives identical results; (2) picture totals are
    the sums of the per-CTU totals; (3) crop invariance: intra decisions only depend on causal neighbours, so the CTUs
    of the top-left 192x128 crop whose neighbourhood is identical in both pictures (row 0: x=0..2, row 1: x=0..1)
    must equal the oracle's result on the crop alone."""
    import oracle
    w, h, bd, qp = 3840, 2160, 10, 32
    planes = synth.frame(w, h, bd, 0, 1234)
    enc = hm.Encoder(w, h, bd, 1, max_batch=2)
    (rec0, ctus0, st0), (rec1, ctus1, st1) = enc.compress([planes, planes], qp)
    # (4) the bitstream pass of the full-size picture (34 substreams per picture, both slots in one launch) equals the oracle's
    # arithmetic coder on the same CTU data byte for byte
    (subs0, nxt0, bins0), (subs1, nxt1, bins1) = enc.encode_slices_run([dict(slice_type=2, qp=qp), dict(slice_type=2, qp=qp)])
    enc.close()
    want_subs, want_nxt, want_bins = oracle.encode_slice(w, h, bd, 1, 2, qp, ctus0)
    assert subs0 == want_subs and subs1 == want_subs and (nxt0, bins0) == (want_nxt, want_bins) and (nxt1, bins1) == (want_nxt, want_bins)
    assert len(subs0) == 34
    common.assert_ctus_equal(ctus0, ctus1, "slot 0 vs slot 1")
    for k in range(3):
        assert np.array_equal(rec0[k], rec1[k])
    assert st0[0] == int(ctus0["total_bits"].astype(np.uint64).sum()) and st0[2] == int(ctus0["total_dist"].astype(np.uint64).sum())
    assert (ctus0["total_bits"] > 0).all()
    cw, ch, wc = 192, 128, (w + 63) // 64
    crop = (planes[0][:ch, :cw].copy(), planes[1][:ch // 2, :cw // 2].copy(), planes[2][:ch // 2, :cw // 2].copy())
    want_rec, want_ctus = oracle.compress(crop, bd, qp, 1)
    full_idx, crop_idx = [0, 1, 2, wc, wc + 1], [0, 1, 2, 3, 4]
    common.assert_ctus_equal(ctus0[full_idx], want_ctus[crop_idx], "4K picture vs oracle on its top-left crop")
    assert np.array_equal(rec0[0][:64, :192], want_rec[0][:64, :192]) and np.array_equal(rec0[0][64:128, :128], want_rec[0][64:128, :128])
    for k in (1, 2):
        assert np.array_equal(rec0[k][:32, :96], want_rec[k][:32, :96]) and np.array_equal(rec0[k][32:64, :64], want_rec[k][32:64, :64])


@pytest.mark.parametrize("name,world,copies,group", [("wpp_416x240_10b_qp32", 2, 1, 1), ("wpp_416x240_10b_qp32", 3, 3, 2), ("wpp_256x192_8b_qp27", 3, 2, 1)])
def test_hip_row_bands_match_unsplit_picture(hm, name, world, copies, group):
    """CTU-row bands: `world` contexts play the ranks of hm-16.2_amd/bands.py's pipeline on one GPU, each searching only its
    band of CTU rows (hm355_run_rows) after importing the row above it from the context that owns it (hm355_export_boundary /
    hm355_import_boundary, host-mediated).  The rows put together equal the reference's unsplit picture: decisions, coefficients,
    costs, reconstruction."""
    import bands
    cfg, frames = common.load_case(name)
    w, h, bd, qp = cfg["width"], cfg["height"], cfg["bit_depth"], cfg["qp"]
    assert cfg["wpp"] == 1
    h_ctu, w_ctu = (h + 63) // 64, (w + 63) // 64
    pics = [(synth.frame(w, h, bd, i % cfg["frames"], cfg["seed"]), frames[i % cfg["frames"]]) for i in range(cfg["frames"] * copies)]
    encs = [hm.Encoder(w, h, bd, 1, max_batch=len(pics)) for _ in range(world)]
    wire = {}

    class Sent:
        def wait(self):
            pass

    for r, enc in enumerate(encs):
        for i, (planes, _) in enumerate(pics):
            enc.upload(i, planes)

        def send(arr, dst, r=r):
            wire.setdefault((r, dst), []).append(np.array(arr, copy=True))
            return Sent()

        def recv(nbytes, src, r=r):
            a = wire[(src, r)].pop(0)
            assert a.size == nbytes
            return a
        bands.run_banded(enc, len(pics), group, h_ctu, r, world, send, recv, qp)      # rank r only needs ranks < r: one after the other
    assert all(not v for v in wire.values())
    if world == 2:
        # the same hand-off with the boundary rows staying on the device (what bench.py --shard rows gives RCCL): rank 1 again, importing
        # from a device buffer that rank 0 exported into
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so.7")               # the HIP runtime the library itself runs on: a plain device allocation
        nb = encs[0].boundary_bytes()
        first1 = bands.band_rows(h_ctu, world, 1)[0]
        dev = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(len(pics) * nb)) == 0
        for i in range(len(pics)):
            encs[0].export_boundary_ptr(i, first1 - 1, dev.value + i * nb)
        back = np.zeros(nb, np.uint8)
        assert hip.hipMemcpy(ctypes.c_void_p(back.ctypes.data), dev, ctypes.c_size_t(nb), 2) == 0     # hipMemcpyDeviceToHost
        assert np.array_equal(back, encs[0].export_boundary(0, first1 - 1))
        enc2 = hm.Encoder(w, h, bd, 1, max_batch=len(pics))
        for i, (planes, _) in enumerate(pics):
            enc2.upload(i, planes)
            enc2.import_boundary_ptr(i, first1 - 1, dev.value + i * nb)
        enc2.run_rows(0, len(pics), qp, first1, h_ctu - 1)
        for i in range(len(pics)):
            a, b = enc2.download(i), encs[1].download(i)
            common.assert_ctus_equal(a[1][first1 * w_ctu:], b[1][first1 * w_ctu:], f"{name} picture {i}: device-resident hand-off")
            assert all(np.array_equal(a[0][k][first1 * (64 >> (1 if k else 0)):], b[0][k][first1 * (64 >> (1 if k else 0)):]) for k in range(3))
        enc2.close()
        hip.hipFree(dev)
    for i, (_, (want_ctus, want_rec)) in enumerate(pics):
        got_ctus = np.zeros_like(encs[0].download(i)[1])
        got_rec = [np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16)]
        for r, enc in enumerate(encs):
            first, last = bands.band_rows(h_ctu, world, r)
            if last < first:
                continue
            rec, ctus, _ = enc.download(i)
            got_ctus[first * w_ctu:(last + 1) * w_ctu] = ctus[first * w_ctu:(last + 1) * w_ctu]
            for k in range(3):
                s = 64 >> (1 if k else 0)
                got_rec[k][first * s:(last + 1) * s] = rec[k][first * s:(last + 1) * s]
        common.assert_ctus_equal(got_ctus, want_ctus, f"{name} picture {i} in {world} bands", (w, h))
        common.assert_rec_equal(got_rec, want_rec, w, h, f"{name} picture {i} in {world} bands")
    for enc in encs:
        enc.close()


@pytest.mark.parametrize("kind,w,h,bd,nref", [("P", 1920, 1080, 8, 4), ("B", 3840, 2160, 10, 2)])
def test_full_size_inter_properties(built, hm, kind, w, h, bd, nref):
    """BASELINE.json configs[2] (encoder_lowdelay_P_main, 1920x1080 8-bit, 4 references, SearchRange 64) and configs[4]
    (encoder_randomaccess_main10 B slice, 3840x2160 10-bit, 2 + 2 references) at full size, WaveFrontSynchro=1, through properties that do
    not need the oracle at that size: (1) the same job in two batch slots gives identical results; (2) picture totals are the sums of
    the per-CTU totals; (3) crop invariance: the four top-left CTUs only read the picture and the reference pictures inside the top-left
    384x256 region (search range 64 + interpolation taps), so they equal the oracle's result on that crop with the cropped references."""
    import math
    import oracle
    qp, seed, cw, ch = 32, 4321, 384, 256
    enc = hm.Encoder(w, h, bd, 1, max_batch=2)
    n = enc.num_ctus

    def crop(planes):
        return (planes[0][:ch, :cw].copy(), planes[1][:ch // 2, :cw // 2].copy(), planes[2][:ch // 2, :cw // 2].copy())

    def motionless(nc):
        mot = np.zeros(nc, [("pred_mode", "u1", 256), ("mv0", "<i2", (256, 2)), ("ref_idx0", "i1", 256), ("mv1", "<i2", (256, 2)), ("ref_idx1", "i1", 256)])
        mot["pred_mode"] = 1; mot["ref_idx0"] = -1; mot["ref_idx1"] = -1
        return mot
    zero = np.zeros((2, 16), np.int32)
    pocs = [3, 2, 1, 0][:nref] if kind == "P" else [0, 8]
    cur_poc = 4
    pics = {poc: synth.frame(w, h, bd, poc, seed) for poc in pocs}          # reference pictures: neighbouring frames of the clip as they are
    finals = {poc: {"poc": poc, "slice_type": 2, "rec": list(pics[poc]), "motion": motionless(n), "num_ref_idx": (0, 0), "ref_poc": zero, "ref_long_term": zero}
              for poc in pocs}
    nc = ((cw + 63) // 64) * ((ch + 63) // 64)
    finals_crop = {poc: dict(f, rec=list(crop(f["rec"])), motion=motionless(nc)) for poc, f in finals.items()}
    ref_poc = np.zeros((2, 16), np.int32)
    ref_poc[0, :len(pocs)] = pocs
    if kind == "B":
        ref_poc[1, :len(pocs)] = pocs[::-1]
    qps = qp + (3 if kind == "P" else 2)
    lam = (0.4624 if kind == "P" else 0.3536) * 2.0 ** ((qps - 12) / 3.0) * min(4.0, max(2.0, (qps - 12) / 6.0))
    srec = {"poc": cur_poc, "slice_type": 1 if kind == "P" else 0, "qp": qps, "lambda": lam, "weight_cb": hm.intra_lambda(qps)[1],
            "cabac_init_type": 1 if kind == "P" else 0, "num_ref_idx": (len(pocs), 0 if kind == "P" else len(pocs)), "ref_poc": ref_poc, "col_from_l0": 1,
            "col_ref_idx": 0, "tmvp": 1, "mvd_l1_zero": 0, "max_merge_cand": 5, "check_ldc": 1 if kind == "P" else 0,
            "lambda_motion_sad": int(math.floor(65536.0 * math.sqrt(lam))), "lambda_motion_sse": int(math.floor(65536.0 * lam))}
    cur = synth.frame(w, h, bd, cur_poc, seed)
    sp, refs = common.ldp_slice_inputs(srec, finals)
    (rec0, ctus0, ictus0, st0), (rec1, ctus1, ictus1, st1) = enc.compress_inter_batch([(cur, sp, refs), (cur, sp, refs)])
    enc.close()
    for f in ctus0.dtype.names:
        assert np.array_equal(ctus0[f], ctus1[f]), f"{kind} {w}x{h}: slot 0 and slot 1 differ in {f}"
    for f in ictus0.dtype.names:
        assert np.array_equal(ictus0[f], ictus1[f]), f"{kind} {w}x{h}: slot 0 and slot 1 differ in {f}"
    for k in range(3):
        assert np.array_equal(rec0[k], rec1[k])
    assert st0[0] == int(ctus0["total_bits"].astype(np.uint64).sum()) and st0[2] == int(ctus0["total_dist"].astype(np.uint64).sum())
    assert (ictus0["inter_dir"] != 0).any() and (kind == "P" or (ictus0["inter_dir"] == 3).any())
    want_rec, want_ctus, want_ictus = oracle.compress_inter(crop(cur), bd, srec, finals_crop, wpp=1)
    wc, wcc = (w + 63) // 64, (cw + 63) // 64
    full_idx, crop_idx = [0, 1, wc, wc + 1], [0, 1, wcc, wcc + 1]
    for f in ("total_bits", "total_dist", "total_cost", "depth", "part_size", "pred_mode", "tr_idx", "cbf", "tskip", "coeff_y", "coeff_cb", "coeff_cr"):
        assert np.array_equal(ctus0[f][full_idx], want_ctus[f][crop_idx]), f"{kind} {w}x{h} vs oracle on the crop: {f} differs"
    for f in ("skip", "merge_flag", "merge_idx", "inter_dir", "mv", "mvd", "ref_idx", "mvp_idx", "mvp_num"):
        assert np.array_equal(ictus0[f][full_idx], want_ictus[f][crop_idx]), f"{kind} {w}x{h} vs oracle on the crop: {f} differs"
    assert np.array_equal(rec0[0][:128, :128], want_rec[0][:128, :128])
    for k in (1, 2):
        assert np.array_equal(rec0[k][:64, :64], want_rec[k][:64, :64])


def test_full_size_1080p_main10_batch_equals_single(hm):
    """BASELINE.json configs[1]: encoder_intra_main10, 1920x1080 10-bit, 8 frames in one batch (WaveFrontSynchro=1 so that a picture
    takes seconds, not a minute) == the same frames searched alone; totals = sums of the per-CTU totals"""
    w, h, bd, qp = 1920, 1080, 10, 32
    frames = [synth.frame(w, h, bd, f, 99) for f in range(8)]
    enc = hm.Encoder(w, h, bd, 1, max_batch=8)
    res = enc.compress(frames, qp)
    for i in (0, 5):
        (rec, ctus, st), = enc.compress([frames[i]], qp)
        common.assert_ctus_equal(ctus, res[i][1], f"1080p frame {i}: alone vs in the batch")
        for k in range(3):
            assert np.array_equal(rec[k], res[i][0][k])
    for rec, ctus, st in res:
        assert st[0] == int(ctus["total_bits"].astype(np.uint64).sum()) and (ctus["total_bits"] > 0).all()
    assert any(not np.array_equal(res[0][1]["depth"], res[i][1]["depth"]) for i in range(1, 8))
    enc.close()


def test_hip_gop_levels_on_two_contexts_match_reference(hm):
    """SURVEY 8e "Inter (C5)": the pictures of one temporal layer on different devices.  Two contexts on one GPU play ranks 0 and 1 of
    hm-16.2_amd/gop_shard.py over the reference's random-access clip (hierarchical GOP of 8): each encodes its share of every level with
    device-resident references (search -> deblocking -> SAO -> hm355_ref_from_slot), exports its finished pictures as blobs into DEVICE buffers
    (hm355_ref_export), the all-gather hands them round and the other context imports them (hm355_ref_import).  Every picture -- decisions,
    motion, coefficients, the finished picture -- equals the reference's single-encoder run; each picture was searched exactly once."""
    import ctypes
    import threading
    import gop_shard
    name = "ra_192x128_10b_qp32"
    saod = {}
    cfg, slices, finals = common.load_ldp_case(name, sao=saod)
    w, h, bd = cfg["width"], cfg["height"], cfg["bit_depth"]
    by_poc = {int(r["poc"]): r for r in slices}
    pics = [dict(poc=int(r["poc"]), depth=saod[int(r["poc"])]["depth"],
                 refs=sorted(set(int(p) for l in range(2) for p in r["ref_poc"][l][:r["num_ref_idx"][l]])) if int(r["slice_type"]) != 2 else []) for r in slices]
    world = 2
    hip = ctypes.CDLL("libamdhip64.so.7")
    barrier = threading.Barrier(world)
    wire, errors, out = [None] * world, [], [None] * world

    class Engine:
        def __init__(self, enc):
            self.enc, self.nb, self.bufs = enc, enc.ref_bytes(), []

        def dev_buffer(self):
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(self.nb)) == 0
            self.bufs.append(p)
            return p.value

        def encode(self, pic, refs, prev_rates):
            r = by_poc[pic["poc"]]
            st, poc = int(r["slice_type"]), pic["poc"]
            planes = synth.frame(w, h, bd, poc, cfg["seed"])
            if st == 2:
                self.enc.upload(0, planes)
                sl = (hm.SliceDesc * 1)(hm.SliceDesc(2, int(r["qp"]), float(r["lambda"]), float(r["weight_cb"])))
                self.enc._check(self.enc.lib.hm355_run(self.enc.h_, 1, sl), "hm355_run")
                _, ctus, _ = self.enc.download(0)
                ictus = None
            else:
                sp, _ = common.ldp_slice_inputs(r, finals)
                _, ctus, ictus, _ = self.enc.compress_inter(planes, sp, {p: refs[p] for p in pic["refs"]})
            self.enc.deblock_run([(st, int(r["qp"]), r["ref_poc"])])
            rate = np.zeros((3, 8), np.float64)
            if pic["depth"] > 0:
                rate[:, pic["depth"] - 1] = prev_rates
            self.enc.sao_run([dict(qp=int(r["qp"]), cabac_init_type=int(r["cabac_init_type"]), depth=pic["depth"], disabled_rate=rate,
                                   chroma_weight=float(r["weight_cb"]), **{"lambda": float(r["lambda"])})])
            fin, _, _ = self.enc.download(0, want_ctus=False)
            handle = self.enc.ref_from_slot(0, poc, st != 2, r["num_ref_idx"], r["ref_poc"], r["ref_long_term"])
            return handle, (ctus, ictus, fin), tuple(rate[:, pic["depth"]])

        def export(self, handle, rates):
            p = self.dev_buffer()
            self.enc.ref_export(handle, p, (*rates, 0.0))
            return p

        def blob_like(self):
            return self.dev_buffer()

        def imp(self, blob):
            handle, user = self.enc.ref_import(int(blob))
            return handle, user[:3]

    def all_gather_for(rank):
        def all_gather(blobs):
            wire[rank] = blobs
            barrier.wait()
            got = [list(b) for b in wire]
            barrier.wait()
            return got
        return all_gather

    def run(rank):
        try:
            enc = hm.Encoder(w, h, bd, cfg["wpp"], max_batch=1)
            eng = Engine(enc)
            res, done = gop_shard.run_gop(eng, pics, rank, world, all_gather_for(rank))
            out[rank] = res
            barrier.wait()                                # nobody frees a buffer the other rank may still be importing from
            for handle, _ in done.values():
                enc.ref_release(handle)
            for p in eng.bufs:
                hip.hipFree(p)
            enc.close()
        except Exception as e:                            # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert sorted(list(out[0]) + list(out[1])) == sorted(by_poc), "every picture exactly once"
    assert len(out[0]) >= 3 and len(out[1]) >= 2
    for res in out:
        for poc, (ctus, ictus, fin) in res.items():
            r = by_poc[poc]
            if ictus is None:
                common.assert_ctus_equal(ctus, common.split_fixture_ctus(r["ctus"])[0], f"{name} POC {poc}")
            else:
                common.assert_inter_ctus_equal(ctus, ictus, r["ctus"], f"{name} POC {poc}")
            for c in range(3):
                assert np.array_equal(fin[c], finals[poc]["rec"][c]), f"{name} POC {poc}: finished picture plane {c}"


def test_row_bands_two_real_ranks_over_gloo():
    """the band pipeline with REAL ranks: two processes (torch.distributed.run, gloo transport) share the GPU, each searches its band of the
    same pictures through hm-16.2_amd/bands.py + TorchTransport, rank 0 gathers the bands and compares them with an unsplit run"""
    import os
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(common.ROOT, "tools", "bands_two_ranks.py"), "416", "240", "3", "2"],
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0 and "bands equal the unsplit run: True" in text, text[-2000:]


def test_create_reports_out_of_memory(hm):
    """a batch that cannot fit the device is refused with HM355_ERR_NOMEM before anything large is allocated"""
    with pytest.raises(RuntimeError, match="rc=-3"):
        hm.Encoder(3840, 2160, 10, 1, max_batch=100000)


def test_hip_primitive_kats(hm):
    """SAD/SSE/SATD and transform kernels vs the reference's known answers"""
    k = np.load(common.GOLD + "/kat_primitives.npz")
    enc = hm.Encoder(64, 64, 8, 0, 1)
    off = k["dist_in_off"]
    for r in range(len(k["dist_out"])):
        v = k["dist_in"][off[r]:off[r + 1]]
        tag = int(v[0])
        if tag == 1:
            bd, n, sub = int(v[1]), int(v[2]), int(v[3]); data = v[4:]
        else:
            bd, n, sub = int(v[1]), int(v[2]), 0; data = v[3:]
        a = data[:n * n].astype(np.int16).reshape(1, n, n); b = data[n * n:].astype(np.int16).reshape(1, n, n)
        kind = {1: (3 if sub else 0), 2: 1, 3: 2}[tag]
        got = int(enc.dist_batch(kind, a, b, bd)[0])
        assert got == int(k["dist_out"][r]), f"record {r} tag {tag} n {n} bd {bd} sub {sub}"
    ioff, ooff = k["tr_in_off"], k["tr_out_off"]
    for r in range(len(ioff) - 1):
        v = k["tr_in"][ioff[r]:ioff[r + 1]]
        tag, bd, n, dst = int(v[0]), int(v[1]), int(v[2]), int(v[3])
        blk = v[4:].astype(np.int32).reshape(1, n, n)
        want = k["tr_out"][ooff[r]:ooff[r + 1]].reshape(n, n)
        got = enc.transform_batch(1 if tag == 5 else 0, blk, bd, dst)[0]
        assert np.array_equal(got, want), f"record {r} tag {tag} n {n} bd {bd}"
    enc.close()


def test_hip_primitives_linearity_at_scale(hm):
    """size-independent property at a large batch: SAD(a,b) == SAD(a-b,0) and SSE symmetric; transform of
    0 is 0 and forward transform is odd: T(-x) == -T(x) up to the rounding offset identity for zero input"""
    rng = np.random.default_rng(3)
    n, count = 16, 4096
    a = rng.integers(0, 1023, (count, n, n)).astype(np.int16); b = rng.integers(0, 1023, (count, n, n)).astype(np.int16)
    enc = hm.Encoder(64, 64, 10, 0, 1)
    s1 = enc.dist_batch(0, a, b, 10); s2 = enc.dist_batch(0, b, a, 10)
    assert np.array_equal(s1, s2)
    ref = (np.abs(a.astype(np.int64) - b).sum(axis=(1, 2)) >> 2).astype(np.uint32)
    assert np.array_equal(s1, ref)
    z = enc.transform_batch(0, np.zeros((8, n, n), np.int32), 10)
    assert not z.any()
    enc.close()


def test_cpp_host_mirror_drop_in(tmp_path):
    """TEncTop::encode -> TEncGOP::compressGOP -> TEncSlice::compressSlice (our C++ mirror of the reference's
    classes) -> C ABI -> HIP: same per-CTU data and reconstruction as the reference fixture"""
    import os, subprocess
    import gen_golden
    cfg, frames = common.load_case("small_128x128_10b_qp37")
    yuv = tmp_path / "in.yuv"
    synth.write_yuv(str(yuv), cfg["width"], cfg["height"], cfg["bit_depth"], cfg["frames"], cfg["seed"])
    dump = tmp_path / "dump.bin"
    exe = os.path.join(common.ROOT, "hm-16.2_amd", "hm355_encmain")
    subprocess.run([exe, str(yuv), str(cfg["width"]), str(cfg["height"]), str(cfg["bit_depth"]), str(cfg["frames"]), str(cfg["qp"]),
                    str(cfg["wpp"]), str(dump)], check=True)
    got = gen_golden.parse_dump(str(dump))
    for i, (ctus, rec) in enumerate(frames):
        common.assert_ctus_equal(got[i][0], ctus, f"frame {i}", (cfg["width"], cfg["height"]))
        assert np.array_equal(got[i][1], rec)
    bits = common.read_mirror_bits(str(dump) + ".bits", len(frames))          # loop filters off: one substream per picture, no SAO syntax
    assert all(len(b) == ((cfg["height"] + 63) // 64 if cfg["wpp"] else 1) and all(len(x) > 0 for x in b) for b in bits)


def test_cpp_host_mirror_with_loop_filters(built, tmp_path):
    """the same C++ mirror with TComLoopFilter::loopFilterPic and TEncSampleAdaptiveOffset::SAOProcess in TEncGOP::compressGOP: the finished
    pictures equal search + deblocking + SAO of the oracle (each pinned against the reference separately; all-intra: temporal depth 0)"""
    import os, subprocess
    import gen_golden, oracle
    w, h, bd, qp, wpp, nf, seed = 200, 136, 10, 30, 1, 2, 9
    yuv = tmp_path / "in.yuv"
    synth.write_yuv(str(yuv), w, h, bd, nf, seed)
    dump = tmp_path / "dump.bin"
    exe = os.path.join(common.ROOT, "hm-16.2_amd", "hm355_encmain")
    subprocess.run([exe, str(yuv), str(w), str(h), str(bd), str(nf), str(qp), str(wpp), str(dump), "lf"], check=True)
    got = gen_golden.parse_dump(str(dump))
    import hm355
    lam, cw = hm355.intra_lambda(qp)
    rate = np.zeros((3, 8), np.float64)
    bits = common.read_mirror_bits(str(dump) + ".bits", nf)
    for i in range(nf):
        planes = synth.frame(w, h, bd, i, seed)
        rec, ctus = oracle.compress(planes, bd, qp, wpp)
        common.assert_ctus_equal(got[i][0], ctus, f"frame {i}")
        dbk = oracle.deblock(rec, bd, qp, 2, np.zeros((2, 16), np.int32), ctus, None)
        fin, params, en = oracle.sao(planes, dbk, bd, qp, lam, cw, 2, 0, rate)
        want = np.concatenate([p.ravel() for p in fin])
        assert np.array_equal(got[i][1], want), f"frame {i}: finished picture differs at {int((got[i][1] != want).sum())} samples"
        # TEncSlice::encodeSlice of the mirror: the slice data (SAO syntax included) equals the oracle's arithmetic coder on the same decisions
        want_subs, _, _ = oracle.encode_slice(w, h, bd, wpp, 2, qp, ctus, sao=params, sao_enabled=(int(en[0]), int(en[1])))
        assert bits[i] == want_subs, f"frame {i}: slice data of the C++ mirror differs"



@pytest.mark.parametrize("name", [common.LDP_CASES[2]] + common.B_CASES + common.LDP_LONG_CASES + ["aq_ldp_256x136_8b_qp32", "aq_ra_192x128_10b_qp30"])     # low-delay P (WPP; two GOPs), random access, low-delay B, AdaptiveQP
def test_cpp_host_mirror_inter_configurations(tmp_path, name):
    """The C++ mirror driven like the reference's encoder on encoder_lowdelay_P_main.cfg (P slices), encoder_lowdelay_main.cfg (B slices,
    list 1 = list 0, mvd_l1_zero, collocated picture from list 1) and encoder_randomaccess_main10.cfg (hierarchical GOP of 8 in coding order,
    references on both sides, the collocated-direction rule): TEncTop::encode queues the GOP, TEncGOP::compressGOP
    derives slice type, QP, lambda, temporal depth, reference list and context table per picture (initEncSlice, the reference picture sets of
    the sequence start, determineCabacInitIdx feedback) and runs search -> deblocking -> SAO -> slice data -> device-resident reference on the
    device.  Everything it derives and everything the device returns must equal the reference's own run of the same clip: slice parameters,
    per-CTU decisions and motion, coefficients, finished pictures and the bytes of every substream."""
    import os, struct, subprocess
    import hmd2
    sd, bd = {}, {}
    cfg, slices, finals = common.load_ldp_case(name, sao=sd, bits=bd)
    w, h = cfg["width"], cfg["height"]
    yuv = tmp_path / "in.yuv"
    synth.write_yuv(str(yuv), w, h, cfg["bit_depth"], cfg["frames"], cfg["seed"])
    dump = tmp_path / "dump.bin"
    exe = os.path.join(common.ROOT, "hm-16.2_amd", "hm355_encmain")
    qp0 = int(slices[0]["qp"])
    subprocess.run([exe, str(yuv), str(w), str(h), str(cfg["bit_depth"]), str(cfg["frames"]), str(qp0), str(cfg["wpp"]), str(dump), "ldb" if name.startswith("ldb") else ("ra" if "ra_" in name else "ldp")] + (["aq"] if name.startswith("aq_") else []), check=True)
    buf = open(dump, "rb").read()
    assert buf[:4] == b"HMD3"
    off = 4 + 20
    bits = common.read_mirror_bits(str(dump) + ".bits", cfg["frames"])
    assert len(slices) == cfg["frames"]
    for i, r in enumerate(slices):
        hdr = struct.unpack_from("<41i", buf, off); off += 164
        lam, = struct.unpack_from("<d", buf, off); off += 8
        n, = struct.unpack_from("<I", buf, off); off += 4
        ctus = np.frombuffer(buf, hmd2.CTU_DT, n, off); off += n * hmd2.CTU_DT.itemsize
        rec = []
        for c in range(3):
            cw, ch = (w, h) if c == 0 else (w // 2, h // 2)
            rec.append(np.frombuffer(buf, "<u2", cw * ch, off).reshape(ch, cw)); off += 2 * cw * ch
        poc, st = int(r["poc"]), int(r["slice_type"])
        nref, nref1 = int(r["num_ref_idx"][0]), int(r["num_ref_idx"][1])
        what = f"{name} POC {poc}"
        assert hdr[:7] == (poc, st, int(r["qp"]), sd[poc]["depth"], int(r["cabac_init_type"]), nref, nref1), f"{what}: slice parameters {hdr[:7]}"
        if st != 2:
            assert hdr[7:9] == (int(r["col_from_l0"]), int(r["mvd_l1_zero"])), f"{what}: collocated list / mvd_l1_zero {hdr[7:9]}"
        assert list(hdr[9:9 + nref]) == [int(v) for v in r["ref_poc"][0][:nref]], f"{what}: list 0 {hdr[9:9 + nref]}"
        assert list(hdr[25:25 + nref1]) == [int(v) for v in r["ref_poc"][1][:nref1]], f"{what}: list 1 {hdr[25:25 + nref1]}"
        assert lam == float(r["lambda"]), f"{what}: lambda {lam} vs {float(r['lambda'])}"
        fields = ["total_cost", "total_bits", "total_dist", "depth", "part_size", "pred_mode", "intra_dir_luma", "intra_dir_chroma", "tr_idx", "cbf", "tskip",
                  "coeff_y", "coeff_cb", "coeff_cr"]
        if st != 2:
            fields += ["skip", "merge_flag", "merge_idx", "inter_dir", "mv0", "mvd0", "ref_idx0", "mvp_idx0", "mvp_num0", "mv1", "mvd1", "ref_idx1", "mvp_idx1", "mvp_num1"]
        for f in fields:
            assert np.array_equal(ctus[f], r["ctus"][f]), f"{what}: {f} differs"
        for c in range(3):
            assert np.array_equal(rec[c], finals[poc]["rec"][c]), f"{what}: finished picture plane {c}"
        assert bits[i] == bd[poc]["substreams"], f"{what}: slice data bytes"
    assert off == len(buf)


def test_hip_pipelined_lanes_equal_blocking_run(hm):
    """hm355_run_begin / hm355_run_wait: four groups of pictures searched by four launches in flight at once (each lane its own stream,
    scratch areas, work list and scheduler words) equal the same pictures through the blocking hm355_run; a lane can be reused after
    its wait, a busy lane and overlapping slots are refused."""
    w, h, bd, qp, per = 192, 128, 10, 30, 3
    planes = [synth.frame(w, h, bd, i % 5, 31) for i in range(4 * per)]
    enc = hm.Encoder(w, h, bd, 1, max_batch=4 * per)
    want = enc.compress(planes, qp)
    for rep in range(2):                                    # the second round reuses every lane's cached work list
        for i, p in enumerate(planes):
            enc.upload(i, planes[(i + rep) % len(planes)])
        for lane in range(4):
            enc.run_begin(lane, lane * per, per, qp)
        with pytest.raises(RuntimeError):
            enc.run_begin(1, per, per, qp)                  # lane 1 is busy
        assert all(enc.run_wait(lane) > 0 for lane in (2, 0, 3, 1))
        with pytest.raises(RuntimeError):
            enc.run_wait(0)                                 # nothing in flight
        for i in range(4 * per):
            rec, ctus, _ = enc.download(i)
            common.assert_ctus_equal(ctus, want[(i + rep) % len(planes)][1], f"round {rep} slot {i}")
            for k in range(3):
                assert np.array_equal(rec[k], want[(i + rep) % len(planes)][0][k])
    enc.run_begin(0, 0, 2 * per, qp)
    with pytest.raises(RuntimeError):
        enc.run_begin(1, per, per, qp)                      # slots overlap the launch in flight on lane 0
    enc.run_wait(0)
    enc.close()
