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
