"""Randomised campaign: the host twin of the kernel source (tests/hostsim/hostsim.cpp, one lane) against the oracle on fresh inputs.
Test infrastructure, CPU only, not collected by pytest.  usage: python tests/fuzz_hostsim.py <worker id> <cases> [path of the built hostsim]
Six kinds of content (the bench's synthetic clip, smooth gradients, full-range noise, screen-like flat blocks with sharp edges, a mix, flat with
outliers), 64..448 x 64..256 samples, 8 and 10 bit, QP 0..51, WPP on and off: every decision array, cost, coefficient and reconstructed sample of every
CTU must be equal.  The twin is built with:  g++ -O2 -std=c++14 -ffp-contract=off -w -o /tmp/hostsim_t tests/hostsim/hostsim.cpp"""
import sys, os, subprocess, time, random, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'hm-16.2_amd'), os.path.join(ROOT, 'oracle'), ROOT]
import numpy as np, common, synth, gen_golden, oracle
wid = int(sys.argv[1]); n = int(sys.argv[2]); EXE = sys.argv[3] if len(sys.argv) > 3 else "/tmp/hostsim_t"; TMP = tempfile.mkdtemp(prefix="hm355_fuzz_")
rng = random.Random(9000 + wid); nrng = np.random.default_rng(9000 + wid)
def content(kind, w, h, bd, seed):
    mx = (1 << bd) - 1
    if kind == 0: return synth.frame(w, h, bd, 0, seed)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    if kind == 1:   # smooth gradients + weak noise: large CUs
        y = ((xx * 3 + yy * 2) // 4 + nrng.integers(0, 3, (h, w))) % (mx + 1)
    elif kind == 2: # strong noise: small CUs, dense coefficients
        y = nrng.integers(0, mx + 1, (h, w))
    elif kind == 3: # screen-like: flat blocks with sharp edges (transform skip, NxN)
        bs = rng.choice([3, 5, 7, 12])
        y = (((xx // bs) * 37 + (yy // bs) * 91) % 7) * (mx // 7)
    elif kind == 4: # mixed: quadrants of the above
        a = content(1, w, h, bd, seed)[0].astype(np.int64); b = content(2, w, h, bd, seed)[0].astype(np.int64); c = content(3, w, h, bd, seed)[0].astype(np.int64)
        y = np.where((xx // 64 + yy // 64) % 3 == 0, a, np.where((xx // 64 + yy // 64) % 3 == 1, (a + b // 8) % (mx + 1), c))
    else:           # flat with a few outliers
        y = np.full((h, w), mx // 2); idx = nrng.integers(0, h * w, 40); y.flat[idx] = nrng.integers(0, mx + 1, 40)
    cy, cx = yy[: h // 2, : w // 2], xx[: h // 2, : w // 2]
    u = (y[::2, ::2] // 2 + cx) % (mx + 1) if kind != 5 else np.full((h // 2, w // 2), mx // 3)
    v = (mx - y[1::2, 1::2] // 3 + cy * 2) % (mx + 1) if kind != 5 else np.full((h // 2, w // 2), mx // 4)
    return y.astype(np.uint16), u.astype(np.uint16), v.astype(np.uint16)
bad = 0
for it in range(n):
    w = rng.randrange(64, 456, 8); h = rng.randrange(64, 264, 8); bd = rng.choice([8, 10]); qp = rng.randrange(0, 52); wpp = rng.choice([0, 1]); seed = rng.randrange(1, 10000); kind = rng.randrange(0, 6)
    planes = content(kind, w, h, bd, seed)
    yuv = os.path.join(TMP, "in.yuv")
    with open(yuv, "wb") as f:
        for p in planes: f.write((p.astype(np.uint8) if bd == 8 else p.astype("<u2")).tobytes())
    t = time.time()
    want_rec, want_ctus = oracle.compress(planes, bd, qp, wpp)
    dump = os.path.join(TMP, "out.bin")
    subprocess.run(["timeout", "600", EXE, yuv, str(w), str(h), str(bd), "1", str(qp), str(wpp), dump], check=True)
    got = gen_golden.parse_dump(dump)
    try:
        common.assert_ctus_equal(got[0][0], want_ctus, f"{w}x{h} bd{bd} qp{qp} wpp{wpp} seed{seed} kind{kind}")
        common.assert_rec_equal(want_rec, got[0][1], w, h, "rec")
        print(wid, it, w, h, bd, qp, wpp, seed, kind, "ok", round(time.time() - t, 1), flush=True)
    except AssertionError as ex:
        bad += 1; print(wid, it, w, h, bd, qp, wpp, seed, kind, "MISMATCH", str(ex)[:200], flush=True)
print(wid, "done, mismatches:", bad, flush=True)
