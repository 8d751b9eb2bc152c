"""diagnostic: the three fresh-input cases of tests/test_gpu_parity.py step by step, with progress lines"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hm-16.2_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import hm355, synth, oracle, common
for (w, h, bd, qp, wpp, seed) in [(192, 128, 8, 30, 1, 11), (128, 72, 10, 34, 0, 12), (64, 64, 8, 25, 0, 13)]:
    planes = synth.frame(w, h, bd, 0, seed)
    want_rec, want_ctus = oracle.compress(planes, bd, qp, wpp)
    print("oracle done", seed, flush=True)
    enc = hm355.Encoder(w, h, bd, wpp, max_batch=1)
    t = time.time()
    (got_rec, got_ctus, _), = enc.compress([planes], qp)
    print("search done", time.time() - t, flush=True)
    common.assert_ctus_equal(got_ctus, want_ctus, f"{w}x{h}")
    print("search equal", flush=True)
    t = time.time()
    try:
        (subs, nxt, bins), = enc.encode_slices_run([dict(slice_type=2, qp=qp)])
        print("bits done", time.time() - t, [len(x) for x in subs], flush=True)
        want_subs, want_nxt, want_bins = oracle.encode_slice(w, h, bd, wpp, 2, qp, want_ctus)
        print("bits equal:", subs == want_subs and nxt == want_nxt and bins == want_bins, flush=True)
    except Exception as ex:
        print("bits failed", time.time() - t, ex, flush=True)
    enc.close()
