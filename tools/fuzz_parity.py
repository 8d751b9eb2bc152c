"""Diagnostic: randomized parity sweep HIP vs oracle (sizes, bit depths, QPs, WPP, slice types, reference counts).
usage: fuzz_parity.py <cases> [seed]      (needs a GPU; the oracle runs on one host core)"""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hm-16.2_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import hm355, oracle, synth, common
cases, seed0 = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed0)
bad = 0
for k in range(cases):
    w, h = int(rng.integers(8, 41)) * 8, int(rng.integers(8, 33)) * 8
    bd, qp, wpp, seed = int(rng.choice([8, 10])), int(rng.integers(18, 41)), int(rng.integers(0, 2)), int(rng.integers(1, 10000))
    kind = str(rng.choice(["I", "P", "B", "B"]))
    t0 = time.time()
    enc = hm355.Encoder(w, h, bd, wpp, max_batch=4)
    try:
        if kind == "I":
            planes = synth.frame(w, h, bd, 0, seed)
            want_rec, want_ctus = oracle.compress(planes, bd, qp, wpp)
            (rec, ctus, _), = enc.compress([planes], qp)
            common.assert_ctus_equal(ctus, want_ctus, "I")
            what = ""
            ictus = None
        else:
            n0, n1 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
            pocs = sorted(set([0, 2, 6, 8][:max(n0, n1) + 1]))
            res = enc.compress([synth.frame(w, h, bd, f, seed) for f in pocs], qp)
            n = enc.num_ctus
            mot = np.zeros(n, [("pred_mode", "u1", 256), ("mv0", "<i2", (256, 2)), ("ref_idx0", "i1", 256), ("mv1", "<i2", (256, 2)), ("ref_idx1", "i1", 256)])
            mot["pred_mode"] = 1; mot["ref_idx0"] = -1; mot["ref_idx1"] = -1
            zero = np.zeros((2, 16), np.int32)
            finals = {poc: {"poc": poc, "slice_type": 2, "rec": res[i][0], "motion": mot, "num_ref_idx": (0, 0), "ref_poc": zero, "ref_long_term": zero}
                      for i, poc in enumerate(pocs)}
            cur_poc = 4
            l0 = sorted(pocs, key=lambda p: (abs(p - cur_poc), p))[:n0]
            l1 = sorted(pocs, key=lambda p: (abs(p - cur_poc), -p))[:n1]
            ref_poc = np.zeros((2, 16), np.int32); ref_poc[0, :len(l0)] = l0
            if kind == "B":
                ref_poc[1, :len(l1)] = l1
            ldc = int(all(p < cur_poc for p in l0 + (l1 if kind == "B" else [])))
            mvd0 = int(kind == "B" and rng.integers(0, 2) == 1 and ldc)
            lam = 0.4624 * 2.0 ** ((qp - 12) / 3.0) * 2.0
            srec = {"poc": cur_poc, "slice_type": 1 if kind == "P" else 0, "qp": qp, "lambda": lam, "weight_cb": hm355.intra_lambda(qp)[1],
                    "cabac_init_type": int(rng.integers(0, 2)), "num_ref_idx": (len(l0), len(l1) if kind == "B" else 0), "ref_poc": ref_poc,
                    "col_from_l0": int(rng.integers(0, 2)) if kind == "B" else 1, "col_ref_idx": 0, "tmvp": int(rng.integers(0, 2)), "mvd_l1_zero": mvd0,
                    "max_merge_cand": int(rng.integers(1, 6)), "check_ldc": ldc,
                    "lambda_motion_sad": int(math.floor(65536.0 * math.sqrt(lam))), "lambda_motion_sse": int(math.floor(65536.0 * lam))}
            cur = synth.frame(w, h, bd, cur_poc, seed)
            want_rec, want_ctus, want_ictus = oracle.compress_inter(cur, bd, srec, finals, wpp=wpp)
            sp, refs = common.ldp_slice_inputs(srec, finals)
            rec, ctus, ictus, _ = enc.compress_inter(cur, sp, refs)
            for f in ("total_bits", "total_dist", "total_cost", "depth", "part_size", "pred_mode", "tr_idx", "cbf", "tskip", "coeff_y", "coeff_cb", "coeff_cr"):
                assert np.array_equal(ctus[f], want_ctus[f]), f
            for f in ("skip", "merge_flag", "merge_idx", "inter_dir", "mv", "mvd", "ref_idx", "mvp_idx", "mvp_num"):
                assert np.array_equal(ictus[f], want_ictus[f]), f
            what = f"L0={l0} L1={l1 if kind == 'B' else []} mrg={srec['max_merge_cand']} tmvp={srec['tmvp']} mvdL1Zero={mvd0} bi={float((ictus['inter_dir'] == 3).mean()):.2f}"
        for c in range(3):
            assert np.array_equal(rec[c], want_rec[c]), f"rec{c}"
        # loop filters on the device (the slot still holds this picture) against the oracle's
        st = {"I": 2, "P": 1, "B": 0}[kind]
        sqp = qp if kind == "I" else int(srec["qp"])
        rp = np.zeros((2, 16), np.int32) if kind == "I" else ref_poc
        lam_s, cw_s = (hm355.intra_lambda(qp) if kind == "I" else (srec["lambda"], srec["weight_cb"]))
        oc = np.zeros(len(ctus), oracle.CTU_DTYPE)
        for f in oc.dtype.names:
            oc[f] = ctus[f]
        oi = None
        if kind != "I":
            oi = np.zeros(len(ctus), oracle.CTU_INTER_DTYPE)
            for f in oi.dtype.names:
                oi[f] = ictus[f]
        want_dbk = oracle.deblock(rec, bd, sqp, st, rp, oc, oi)
        depth = int(rng.integers(0, 3)); rate = rng.random((3, 8)); rate_o = rate.copy()
        cit = 2 if kind == "I" else int(srec["cabac_init_type"])
        cur_planes = planes if kind == "I" else cur
        want_sao, want_par, want_en = oracle.sao(cur_planes, want_dbk, bd, sqp, lam_s, cw_s, cit, depth, rate_o)
        enc.deblock_run([(st, sqp, rp)])
        got_dbk, _, _ = enc.download(0, want_ctus=False)
        for c in range(3):
            assert np.array_equal(got_dbk[c], want_dbk[c]), f"deblocked plane {c}"
        (en, par), = enc.sao_run([dict(qp=sqp, cabac_init_type=cit, depth=depth, disabled_rate=rate, chroma_weight=cw_s, **{"lambda": lam_s})])
        got_sao, _, _ = enc.download(0, want_ctus=False)
        assert tuple(en) == tuple(int(v) for v in want_en), "SAO slice flags"
        assert np.array_equal(common.normalise_sao(par), common.normalise_sao(want_par)), "SAO parameters"
        assert np.allclose(rate, rate_o), "SAO disabled rates"
        for c in range(3):
            assert np.array_equal(got_sao[c], want_sao[c]), f"SAO output plane {c}"
        what += f" sao(depth {depth}, en {tuple(en)}, new {int((par[:, :, 0] == 1).sum())}, merge {int((par[:, :, 0] == 2).sum())})"
        # bitstream pass on the same slot (CU data, coefficients, SAO parameters resident) against the oracle's arithmetic coder
        hdr = {} if kind == "I" else dict(cabac_init_type=int(srec["cabac_init_type"]), num_ref_idx=srec["num_ref_idx"], mvd_l1_zero=int(srec["mvd_l1_zero"]),
                                          max_merge_cand=int(srec["max_merge_cand"]))
        (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=st, qp=sqp, sao_enabled=(en[0], en[1]), **hdr)])
        want_subs, want_nxt, want_bins = oracle.encode_slice(w, h, bd, wpp, st, sqp, oc, oi, sao=want_par, sao_enabled=(int(want_en[0]), int(want_en[1])), **hdr)
        assert subs == want_subs, "slice data bytes"
        assert (nxt, bins) == (want_nxt, want_bins), "next context table / bin count"
        what += f" bits({sum(len(x) for x in subs)} B, {len(subs)} substreams, next table {nxt})"
        print(f"case {k}: {kind} {w}x{h} {bd}b qp{qp} wpp{wpp} seed{seed} {what} ok ({time.time() - t0:.1f}s)", flush=True)
    except AssertionError as ex:
        bad += 1
        print(f"case {k}: {kind} {w}x{h} {bd}b qp{qp} wpp{wpp} seed{seed} MISMATCH {str(ex)[:120]}", flush=True)
    enc.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
