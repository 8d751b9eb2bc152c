"""Seeded, integer-only synthetic 4:2:0 clips.

Every sample is a pure function of (x, y, frame, seed) in uint32 arithmetic, so the
same clip can be regenerated on any box without a data file.  The picture mixes
  * smooth triangle-wave gradients that translate by (3, -2) px/frame,
  * xorshift-hashed texture whose amplitude changes per 64x64 / 32x32 region
    (flat regions -> large CUs, busy regions -> small CUs at QP 32),
  * hard-edged rectangles (one moving 7,5 px/frame) and a diagonal edge,
so that the HM CU quadtree uses all depths 0..3.
Planar output: Y (h x w), Cb, Cr (h/2 x w/2) as uint16 arrays holding bit_depth-bit samples.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _hash32(x, y, f, seed):
    h = (x.astype(np.uint64) * np.uint64(73856093)) ^ (y.astype(np.uint64) * np.uint64(19349663))
    h ^= np.uint64((f * 83492791 + seed * 2654435761) & 0xFFFFFFFF)
    h &= _M32
    for _ in range(2):
        h ^= (h << np.uint64(13)) & _M32
        h ^= h >> np.uint64(17)
        h ^= (h << np.uint64(5)) & _M32
    return h.astype(np.int64)


def _tri(v, period, amp):
    # integer triangle wave in [0, amp]
    p = np.mod(v, 2 * period)
    t = np.where(p < period, p, 2 * period - p)
    return (t * amp) // period


def frame(width, height, bit_depth, f, seed=1234):
    """Return (Y, Cb, Cr) uint16 planes of frame f."""
    yy, xx = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    # ---- luma, computed at 8-bit precision then widened ----
    base = 60 + _tri(xx + 3 * f, 157, 70) + _tri(yy - 2 * f + 4096, 211, 50)
    # per-region texture strength: 0, 4, 12, 28, 60 (amplitude mask) on a 64x64 grid refined by 32x32 parity
    region = ((xx >> 6) * 5 + (yy >> 6) * 3 + ((xx >> 5) & 1) * ((yy >> 5) & 1) * 2 + (seed & 7)) % 5
    mask = (4 << region) - 4                  # 0,4,12,28,60
    h = _hash32(xx + 3 * f, yy - 2 * f + 4096, 0, seed)
    noise = (h & mask) - (mask >> 1)
    # a textured 16x16 checker in some regions -> mid-size CUs
    chk = (((xx >> 4) + (yy >> 4)) & 1) * ((region == 2) * 12)
    img = base + noise + chk
    # moving bright rectangle, 128x96
    bx = (40 + 7 * f) % max(1, width - 128)
    by = (24 + 5 * f) % max(1, height - 96)
    inside = (xx >= bx) & (xx < bx + 128) & (yy >= by) & (yy < by + 96)
    img = np.where(inside, img + 45, img)
    # static dark diagonal band
    band = ((xx + 2 * yy) % 389) < 23
    img = np.where(band, img - 35, img)
    img = np.clip(img, 0, 255)
    shift = bit_depth - 8
    if shift > 0:
        low = _hash32(xx, yy, f + 17, seed) >> 7 & ((1 << shift) - 1)
        Y = (img << shift) + low
    else:
        Y = img
    # ---- chroma: smooth ramps + weak noise, a coloured rectangle ----
    cy, cx = np.meshgrid(np.arange(height // 2, dtype=np.int64), np.arange(width // 2, dtype=np.int64), indexing="ij")
    hc = _hash32(cx, cy, f + 101, seed)
    cb = 110 + _tri(cx + f, 97, 36) + ((hc & 6) - 3)
    cr = 140 - _tri(cy + 2 * f, 131, 40) + (((hc >> 8) & 6) - 3)
    cin = (cx * 2 >= bx) & (cx * 2 < bx + 128) & (cy * 2 >= by) & (cy * 2 < by + 96)
    cb = np.where(cin, cb + 20, cb)
    cr = np.where(cin, cr - 25, cr)
    cb = np.clip(cb, 0, 255)
    cr = np.clip(cr, 0, 255)
    if shift > 0:
        cb = (cb << shift) + ((hc >> 16) & ((1 << shift) - 1))
        cr = (cr << shift) + ((hc >> 20) & ((1 << shift) - 1))
    return Y.astype(np.uint16), cb.astype(np.uint16), cr.astype(np.uint16)


def write_yuv(path, width, height, bit_depth, n_frames, seed=1234):
    """Write the clip as planar 4:2:0 (uint8 for 8-bit, little-endian uint16 otherwise)."""
    with open(path, "wb") as fh:
        for f in range(n_frames):
            for p in frame(width, height, bit_depth, f, seed):
                if bit_depth == 8:
                    fh.write(p.astype(np.uint8).tobytes())
                else:
                    fh.write(p.astype("<u2").tobytes())
