// hm355 -- inter (P / B slice) part of the CTU search: merge / AMVP / temporal candidates, TZ integer search, fractional
// refinement, motion compensation, inter residual quadtree and inter syntax.  Included by hm355_core.h; same execution
// model (one wavefront per CTU, wave-uniform decisions, lane-parallel sample work).  Reference file:line as in the rest.
#pragma once

#define SIZE_2NxN 1
#define SIZE_Nx2N 2
#define SIZE_2NxnU 4
#define SIZE_2NxnD 5
#define SIZE_nLx2N 6
#define SIZE_nRx2N 7
#define MODE_INTER 0
#define HM_P_SLICE 1
#define HM_B_SLICE 0

// ------------------------------------------------------------------------------------------------
// interpolation (TComInterpolationFilter.cpp:55-260) and motion compensation (TComPrediction.cpp:586-697)
// ------------------------------------------------------------------------------------------------
HM_CONST int8_t HM_LUMA_FILTER[4][8] = { {0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1} };
HM_CONST int8_t HM_CHROMA_FILTER[8][4] = { {0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2} };
#define HM_IF_PREC 14
#define HM_IF_OFFS (1 << (HM_IF_PREC - 1))

// one output sample of TComInterpolationFilter::filter<N, isVertical, isFirst, isLast> / filterCopy; src points at the
// sample the filter is centred on (tap N/2-1), cs = distance between taps
HM_DEV inline Pel if_sample(int bitDepth, const Pel *src, int cs, int frac, int chroma, int isFirst, int isLast)
{
  const int headRoom = (HM_IF_PREC - bitDepth) > 2 ? (HM_IF_PREC - bitDepth) : 2;
  if (frac == 0) { // filterCopy :86-150
    if (isFirst == isLast) return src[0];
    if (isFirst) return (Pel)((Pel)(src[0] << headRoom) - (Pel)HM_IF_OFFS);
    Pel v = (Pel)((src[0] + HM_IF_OFFS + (1 << (headRoom - 1))) >> headRoom);
    const int maxv = (1 << bitDepth) - 1;
    return v < 0 ? (Pel)0 : (v > maxv ? (Pel)maxv : v);
  }
  int sum = 0;
  if (!chroma) { const Pel *p = src - 3 * cs;
#pragma unroll
    for (int t = 0; t < 8; t++) sum += p[t * cs] * HM_LUMA_FILTER[frac][t]; }
  else { const Pel *p = src - cs;
#pragma unroll
    for (int t = 0; t < 4; t++) sum += p[t * cs] * HM_CHROMA_FILTER[frac][t]; }
  int shift = 6, offset;
  if (isLast) { shift += isFirst ? 0 : headRoom; offset = 1 << (shift - 1); offset += isFirst ? 0 : HM_IF_OFFS << 6; }
  else { shift -= isFirst ? headRoom : 0; offset = isFirst ? -HM_IF_OFFS << shift : 0; }
  Pel val = (Pel)((sum + offset) >> shift);
  if (isLast) { const int maxv = (1 << bitDepth) - 1; val = val < 0 ? (Pel)0 : (val > maxv ? (Pel)maxv : val); }
  return val;
}
// block prediction at a (possibly fractional) displacement; r = reference sample of the block's top-left at the integer
// part of the motion vector; cw x ch samples; always two-stage when twoStage != 0 (the fractional search planes)
HM_DEV inline void interp_block(Shared *e, int chroma, const Pel *r, int refStride, int xFrac, int yFrac, int cw, int ch, Pel *dst, int dstStride, int twoStage, int bi = 0)
{
  const int bd = e->bitDepth, last = !bi;
  if (!twoStage && yFrac == 0) { HM_PAR_FOR_XY(x, y, cw, cw * ch) dst[y * dstStride + x] = if_sample(bd, r + y * refStride + x, 1, xFrac, chroma, 1, last); HM_SYNC(); return; }
  if (!twoStage && xFrac == 0) { HM_PAR_FOR_XY(x, y, cw, cw * ch) dst[y * dstStride + x] = if_sample(bd, r + y * refStride + x, refStride, yFrac, chroma, 1, last); HM_SYNC(); return; }
  const int half = chroma ? 1 : 3, rows = ch + (chroma ? 3 : 7);
  const int inLds = cw * rows <= (int)(sizeof(e->bufA) / sizeof(Pel));         // intermediate rows in LDS (idle transform buffer) when they fit
  Pel *tmp = inLds ? (Pel *)e->bufA : e->ws->mcTmp; const int ts = inLds ? cw : 64;
  HM_PAR_FOR_XY(x, y, cw, cw * rows) tmp[y * ts + x] = if_sample(bd, r + (y - half) * refStride + x, 1, xFrac, chroma, 1, 0);
  HM_SYNC();
  HM_PAR_FOR_XY(x, y, cw, cw * ch) dst[y * dstStride + x] = if_sample(bd, tmp + (y + half) * ts + x, ts, yFrac, chroma, 0, last);
  HM_SYNC();
}
HM_DEV inline MvD clip_mv(const Shared *e, MvD mv, int cuX, int cuY)
{ // TComDataCU::clipMv, TComDataCU.cpp:2917-2932
  const int off = 8;
  const int horMax = (e->width + off - cuX - 1) << 2, horMin = (-64 - off - cuX + 1) << 2;
  const int verMax = (e->height + off - cuY - 1) << 2, verMin = (-64 - off - cuY + 1) << 2;
  int x = mv.x, y = mv.y;
  x = x < horMin ? horMin : x; x = x > horMax ? horMax : x;
  y = y < verMin ? verMin : y; y = y > verMax ? verMax : y;
  MvD r; r.x = (int16_t)x; r.y = (int16_t)y;
  return r;
}
// TComPrediction::xPredInterBlk :660-697 (uni-directional): (px,py) luma position in the picture, w x h luma size
HM_DEV inline void pred_inter_blk(Shared *e, int comp, const RefPicDev *ref, int px, int py, MvD mv, int w, int h, Pel *dst, int dstStride, int bi = 0)
{
  const int sh = comp ? 1 : 0, shift = 2 + sh;
  const int refStride = ref->stride[comp];
  const Pel *r = ref->plane[comp] + (ptrdiff_t)((py >> sh) + (mv.y >> shift)) * refStride + (px >> sh) + (mv.x >> shift);
  interp_block(e, comp != 0, r, refStride, mv.x & ((1 << shift) - 1), mv.y & ((1 << shift) - 1), w >> sh, h >> sh, dst, dstStride, 0, bi);
}

// ------------------------------------------------------------------------------------------------
// rectangular distortion (AMP shapes): SAD with row sub-sampling, SATD by 8x8 / 4x4 blocks
// ------------------------------------------------------------------------------------------------
HM_DEV inline uint32_t dist_sad_rect(const Pel *org, int so, const Pel *cur, int sc, int w, int h, int subShift, int bitDepth)
{
  const int rows = h >> subShift;
  uint32_t sum = 0;
  const int so2 = so << subShift, sc2 = sc << subShift;
  HM_PAR_FOR_XY(x, yy, w, rows * w) sum += (uint32_t)hm_abs(org[yy * so2 + x] - cur[yy * sc2 + x]);
  return (hm_wave_sum(sum) << subShift) >> (bitDepth - 8);
}
#if !defined(HM355_HOSTSIM)
// Small blocks leave most lanes idle when a lane owns a whole 8x8 / 4x4 Hadamard block, so they are transformed across lanes
// instead: a lane holds one difference sample and the 2-D Hadamard is log2(N) butterfly exchange stages (the set of |coefficients|
// does not depend on the butterfly order, so the sum equals xCalcHADs8x8 / xCalcHADs4x4 exactly).
__device__ __forceinline__ int hm_bfly(int v, int m)
{
  int t;
  if (m == 1) t = hm_dpp<0xB1>(0, v);                               // quad_perm [1,0,3,2]
  else if (m == 2) t = hm_dpp<0x4E>(0, v);                          // quad_perm [2,3,0,1]
  else t = __shfl_xor(v, m, 64);
  return (hm_lane() & m) ? (t - v) : (v + t);
}
__device__ __forceinline__ uint32_t had8_wave(const Pel *org, int so, const Pel *cur, int sc)
{ // one 8x8 block per pass: lane = y*8 + x
  const int l = hm_lane(), x = l & 7, y = l >> 3;
  int v = org[y * so + x] - cur[y * sc + x];
  v = hm_bfly(v, 1); v = hm_bfly(v, 2); v = hm_bfly(v, 4); v = hm_bfly(v, 8); v = hm_bfly(v, 16); v = hm_bfly(v, 32);
  return (hm_wave_sum((uint32_t)hm_abs(v)) + 2) >> 2;
}
__device__ __forceinline__ uint32_t had4_wave4(const Pel *org, int so, const Pel *cur, int sc, int b0, int nb, int nbx)
{ // four 4x4 blocks per pass, one per 16-lane row: lane = blk*16 + y*4 + x
  const int l = hm_lane(), bi = b0 + (l >> 4), x = l & 3, y = (l >> 2) & 3;
  int v = 0;
  if (bi < nb) { const int by = bi / nbx, bx = bi - by * nbx; v = org[(by * 4 + y) * so + bx * 4 + x] - cur[(by * 4 + y) * sc + bx * 4 + x]; }
  v = hm_bfly(v, 1); v = hm_bfly(v, 2); v = hm_bfly(v, 4); v = hm_bfly(v, 8);
  int a = hm_abs(v);
  a += hm_dpp<0x111>(0, a); a += hm_dpp<0x112>(0, a); a += hm_dpp<0x114>(0, a); a += hm_dpp<0x118>(0, a);     // per-row sums in lanes 15/31/47/63
  const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane(a, 15), s1 = (uint32_t)__builtin_amdgcn_readlane(a, 31);
  const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane(a, 47), s3 = (uint32_t)__builtin_amdgcn_readlane(a, 63);
  return ((s0 + 1) >> 1) + ((s1 + 1) >> 1) + ((s2 + 1) >> 1) + ((s3 + 1) >> 1);
}
#endif
HM_DEV inline uint32_t dist_hads_rect(const Pel *org, int so, const Pel *cur, int sc, int w, int h, int bitDepth)
{ // xGetHADs, TComRdCost.cpp:1537-1606
  uint32_t sum = 0;
  if ((w & 7) == 0 && (h & 7) == 0) {
    const int nbx = w >> 3, nb = nbx * (h >> 3);
#if !defined(HM355_HOSTSIM)
    if (nb <= 12) {
      for (int b = 0; b < nb; b++) { const int by = b / nbx, bx = b - by * nbx; sum += had8_wave(org + by * 8 * so + bx * 8, so, cur + by * 8 * sc + bx * 8, sc); }
      return sum >> (bitDepth - 8);
    }
#endif
    HM_PAR_FOR(b, nb) { const int by = b / nbx, bx = b - by * nbx; sum += had8(org + by * 8 * so + bx * 8, so, cur + by * 8 * sc + bx * 8, sc); }
  } else {
    const int nbx = w >> 2, nb = nbx * (h >> 2);
#if !defined(HM355_HOSTSIM)
    for (int b0 = 0; b0 < nb; b0 += 4) sum += had4_wave4(org, so, cur, sc, b0, nb, nbx);
    return sum >> (bitDepth - 8);
#endif
    HM_PAR_FOR(b, nb) { const int by = b / nbx, bx = b - by * nbx; sum += had4(org + by * 4 * so + bx * 4, so, cur + by * 4 * sc + bx * 4, sc); }
  }
  return hm_wave_sum(sum) >> (bitDepth - 8);
}

// ------------------------------------------------------------------------------------------------
// partition geometry (TComDataCU::getPartIndexAndSize :1993)
// ------------------------------------------------------------------------------------------------
struct Rect { int x, y, w, h; };       // luma samples, relative to the CTU
HM_DEV inline int num_parts_of(int partSize) { return partSize == SIZE_2Nx2N ? 1 : (partSize == SIZE_NxN ? 4 : 2); }
HM_DEV inline Rect pu_rect(int cuZ, int cuDepth, int partSize, int partIdx)
{
  const int n = 64 >> cuDepth, rz = hm_z2r(cuZ);
  Rect r; r.x = (rz & 15) * 4; r.y = (rz >> 4) * 4; r.w = n; r.h = n;
  switch (partSize) {
    case SIZE_2NxN:  r.h = n >> 1; if (partIdx) r.y += n >> 1; break;
    case SIZE_Nx2N:  r.w = n >> 1; if (partIdx) r.x += n >> 1; break;
    case SIZE_NxN:   r.w = r.h = n >> 1; r.x += (partIdx & 1) * (n >> 1); r.y += (partIdx >> 1) * (n >> 1); break;
    case SIZE_2NxnU: if (partIdx == 0) r.h = n >> 2; else { r.y += n >> 2; r.h = (n >> 2) + (n >> 1); } break;
    case SIZE_2NxnD: if (partIdx == 0) r.h = (n >> 2) + (n >> 1); else { r.y += (n >> 2) + (n >> 1); r.h = n >> 2; } break;
    case SIZE_nLx2N: if (partIdx == 0) r.w = n >> 2; else { r.x += n >> 2; r.w = (n >> 2) + (n >> 1); } break;
    case SIZE_nRx2N: if (partIdx == 0) r.w = (n >> 2) + (n >> 1); else { r.x += (n >> 2) + (n >> 1); r.w = n >> 2; } break;
    default: break;
  }
  return r;
}
HM_DEV inline int rect_z(Rect r) { return hm_r2z((r.y >> 2) * 16 + (r.x >> 2)); }
// lane-parallel loop over the 4x4 partitions of a PU
#define HM_PU_FOR(r, z) HM_PAR_FOR(pi_, ((r).w >> 2) * ((r).h >> 2)) for (int z = hm_r2z((((r).y >> 2) + pi_ / ((r).w >> 2)) * 16 + ((r).x >> 2) + pi_ % ((r).w >> 2)), once_ = 1; once_; once_ = 0)

HM_DEV inline void pu_set_motion(Shared *e, Rect r, int list, MvD mv, int refIdx)
{ InterMeta *m = e->im; HM_PU_FOR(r, z) { m->mv[list][z] = mv; m->refIdx[list][z] = (int8_t)refIdx; } HM_SYNC(); }
HM_DEV inline void pu_set_mvd(Shared *e, Rect r, int list, MvD mvd) { InterMeta *m = e->im; HM_PU_FOR(r, z) m->mvd[list][z] = mvd; HM_SYNC(); }
HM_DEV inline void pu_set_mvp(Shared *e, Rect r, int list, int idx, int num) { InterMeta *m = e->im; HM_PU_FOR(r, z) { m->mvpIdx[list][z] = (int8_t)idx; m->mvpNum[list][z] = (int8_t)num; } HM_SYNC(); }
HM_DEV inline void pu_set_u8(uint8_t *arr, Rect r, int v) { HM_PU_FOR(r, z) arr[z] = (uint8_t)v; HM_SYNC(); }

// ------------------------------------------------------------------------------------------------
// neighbour access (TComDataCU::getPULeft/Above/AboveLeft/AboveRight/BelowLeft, TComDataCU.cpp:1043-1290)
// ------------------------------------------------------------------------------------------------
enum { NB_LEFT, NB_ABOVE, NB_ABOVE_LEFT, NB_ABOVE_RIGHT, NB_BELOW_LEFT };
// returns the raster address of the CTU holding the neighbour (or -1) and its z index
HM_DEV inline int nb_at(const Shared *e, int x4, int y4, int dir, int curZ, int *z)
{
  int nx = x4, ny = y4;
  switch (dir) {
    case NB_LEFT: nx--; break;
    case NB_ABOVE: ny--; break;
    case NB_ABOVE_LEFT: nx--; ny--; break;
    case NB_ABOVE_RIGHT: nx++; ny--; break;
    default: nx--; ny++; break;
  }
  const int gx = e->ctuX * 16 + nx, gy = e->ctuY * 16 + ny;
  if (gx < 0 || gy < 0) return -1;
  if (gx * 4 >= e->width || gy * 4 >= e->height) return -1;
  const int cx = gx >> 4, cy = gy >> 4;
  if (cy > e->ctuY) return -1;
  if (cy == e->ctuY && cx > e->ctuX) return -1;
  if (cy < e->ctuY && cx > e->ctuX + 1) return -1;
  *z = hm_r2z((gy & 15) * 16 + (gx & 15));
  if (cx == e->ctuX && cy == e->ctuY && (dir == NB_ABOVE_RIGHT || dir == NB_BELOW_LEFT) && !(curZ > *z)) return -1;
  return cy * e->wCtu + cx;
}
HM_DEV inline const CtuMeta *cmeta_of(const Shared *e, int ca) { return ca == e->ctuAddr ? &e->meta : e->fb.meta + ca; }
HM_DEV inline const InterMeta *imeta_of(const Shared *e, int ca) { return ca == e->ctuAddr ? e->im : e->fb.imeta + ca; }   // (a team helper searches on a private copy, hm355_team.h)
HM_DEV inline int nb_is_inter(const Shared *e, int ca, int z) { return cmeta_of(e, ca)->pred[z] == MODE_INTER; }

// ------------------------------------------------------------------------------------------------
// temporal candidate (xGetColMVP :3196, xGetDistScaleFactor :3290)
// ------------------------------------------------------------------------------------------------
HM_DEV inline int dist_scale_factor(int currPOC, int currRefPOC, int colPOC, int colRefPOC)
{
  const int diffD = colPOC - colRefPOC, diffB = currPOC - currRefPOC;
  if (diffD == diffB) return 4096;
  const int tdb = hm_clip3(-128, 127, diffB), tdd = hm_clip3(-128, 127, diffD);
  const int x = (0x4000 + hm_abs(tdd / 2)) / tdd;
  return hm_clip3(-4096, 4095, (tdb * x + 32) >> 6);
}
HM_DEV inline MvD scale_mv(MvD mv, int scale)
{
  MvD r;
  r.x = (int16_t)hm_clip3(-32768, 32767, (scale * mv.x + 127 + (scale * mv.x < 0)) >> 8);
  r.y = (int16_t)hm_clip3(-32768, 32767, (scale * mv.y + 127 + (scale * mv.y < 0)) >> 8);
  return r;
}
HM_DEV inline int get_col_mvp(const Shared *e, int list, int ctuAddr, int z, MvD *out, int refIdx)
{
  const InterPic *s = e->fb.ip;
  const RefPicDev *col = &s->ref[s->sliceType == HM_B_SLICE ? 1 - s->colFromL0 : 0][s->colRefIdx];
  const size_t p = (size_t)ctuAddr * 256 + z;
  if (col->predMode[p] != MODE_INTER) return 0;
  int colList = s->checkLDC ? list : s->colFromL0;
  int colRefIdx = col->refIdx[colList][p];
  if (colRefIdx < 0) { colList = 1 - colList; colRefIdx = col->refIdx[colList][p]; if (colRefIdx < 0) return 0; }
  const int colRefPOC = col->refPoc[colList][colRefIdx];
  const MvD colMv = col->mv[colList][p];
  const RefPicDev *cur = &s->ref[list][refIdx];
  const int curLT = cur->isLongTerm, colLT = col->refLT[colList][colRefIdx];
  if (curLT != colLT) return 0;
  if (curLT || colLT) *out = colMv;
  else { const int scale = dist_scale_factor(s->poc, cur->poc, col->poc, colRefPOC); *out = scale == 4096 ? colMv : scale_mv(colMv, scale); }
  return 1;
}
HM_DEV inline int temporal_mv(const Shared *e, Rect r, int list, int refIdx, MvD *out)
{
  const int rbx = (r.x + r.w - 4) >> 2, rby = (r.y + r.h - 4) >> 2;
  int ctuAddr = -1, z = 0;
  if ((e->ctuX * 64 + rbx * 4 + 4) < e->width && (e->ctuY * 64 + rby * 4 + 4) < e->height) {
    if (rbx < 15 && rby < 15) { z = hm_r2z((rby + 1) * 16 + rbx + 1); ctuAddr = e->ctuAddr; }
    else if (rbx < 15) { }
    else if (rby < 15) { z = hm_r2z((rby + 1) * 16); ctuAddr = e->ctuAddr + 1; }
  }
  if (ctuAddr >= 0 && get_col_mvp(e, list, ctuAddr, z, out, refIdx)) return 1;
  const int cx = (r.x >> 2) + ((r.w >> 2) / 2), cy = (r.y >> 2) + ((r.h >> 2) / 2);
  return get_col_mvp(e, list, e->ctuAddr, hm_r2z(cy * 16 + cx), out, refIdx);
}

// ------------------------------------------------------------------------------------------------
// merge candidates (getInterMergeCandidates :2309-2662)
// ------------------------------------------------------------------------------------------------
HM_DEV inline int equal_motion(const Shared *e, int ca, int za, int cb, int zb)
{ // hasEqualMotion :2278
  const InterMeta *a = imeta_of(e, ca), *b = imeta_of(e, cb);
  if (a->interDir[za] != b->interDir[zb]) return 0;
  for (int l = 0; l < 2; l++)
    if (a->interDir[za] & (1 << l))
      if (a->mv[l][za].x != b->mv[l][zb].x || a->mv[l][za].y != b->mv[l][zb].y || a->refIdx[l][za] != b->refIdx[l][zb]) return 0;
  return 1;
}
HM_DEV inline void merge_take(const Shared *e, MergeList *ml, int cnt, int ca, int z, int isB)
{
  const InterMeta *m = imeta_of(e, ca); ml->dir[cnt] = m->interDir[z]; ml->f[cnt][0].mv = m->mv[0][z]; ml->f[cnt][0].ref = m->refIdx[0][z];
  if (isB) { ml->f[cnt][1].mv = m->mv[1][z]; ml->f[cnt][1].ref = m->refIdx[1][z]; }
}
HM_DEV HM_NOINLINE void merge_candidates(Shared *e, int cuZ, int cuDepth, int partSize, int puIdx, MergeList *ml)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); puIdx = HM_UNI(puIdx); ml = hm_uni_ptr(ml);
  const InterPic *s = e->fb.ip; const int maxC = s->maxMergeCand, isB = s->sliceType == HM_B_SLICE;
  const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
  for (int i = 0; i < 5; i++) { ml->dir[i] = 0; for (int l = 0; l < 2; l++) { ml->f[i][l].mv.x = ml->f[i][l].mv.y = 0; ml->f[i][l].ref = -1; } }
  ml->num = maxC;
  int cnt = 0;
  const int ltx = r.x >> 2, lty = r.y >> 2, rtx = (r.x + r.w - 4) >> 2, lbx = ltx, lby = (r.y + r.h - 4) >> 2;
  const int zLT = hm_r2z(lty * 16 + ltx), zRT = hm_r2z(lty * 16 + rtx), zLB = hm_r2z(lby * 16 + lbx);
  int zl = 0, za = 0, zt = 0;
  const int cL = nb_at(e, lbx, lby, NB_LEFT, zLB, &zl);
  const int availA1 = cL >= 0 && !(puIdx == 1 && (partSize == SIZE_Nx2N || partSize == SIZE_nLx2N || partSize == SIZE_nRx2N)) && nb_is_inter(e, cL, zl);
  if (availA1) { merge_take(e, ml, cnt, cL, zl, isB); cnt++; }
  if (cnt == maxC) return;
  const int cA = nb_at(e, rtx, lty, NB_ABOVE, zRT, &za);
  const int availB1 = cA >= 0 && !(puIdx == 1 && (partSize == SIZE_2NxN || partSize == SIZE_2NxnU || partSize == SIZE_2NxnD)) && nb_is_inter(e, cA, za);
  if (availB1 && (!availA1 || !equal_motion(e, cL, zl, cA, za))) { merge_take(e, ml, cnt, cA, za, isB); cnt++; }
  if (cnt == maxC) return;
  int cT = nb_at(e, rtx, lty, NB_ABOVE_RIGHT, zRT, &zt);
  const int availB0 = cT >= 0 && nb_is_inter(e, cT, zt);
  if (availB0 && (!availB1 || !equal_motion(e, cA, za, cT, zt))) { merge_take(e, ml, cnt, cT, zt, isB); cnt++; }
  if (cnt == maxC) return;
  cT = nb_at(e, lbx, lby, NB_BELOW_LEFT, zLB, &zt);
  const int availA0 = cT >= 0 && nb_is_inter(e, cT, zt);
  if (availA0 && (!availA1 || !equal_motion(e, cL, zl, cT, zt))) { merge_take(e, ml, cnt, cT, zt, isB); cnt++; }
  if (cnt == maxC) return;
  if (cnt < 4) {
    cT = nb_at(e, ltx, lty, NB_ABOVE_LEFT, zLT, &zt);
    const int availB2 = cT >= 0 && nb_is_inter(e, cT, zt);
    if (availB2 && (!availA1 || !equal_motion(e, cL, zl, cT, zt)) && (!availB1 || !equal_motion(e, cA, za, cT, zt))) { merge_take(e, ml, cnt, cT, zt, isB); cnt++; }
  }
  if (cnt == maxC) return;
  if (s->tmvp) {
    int dir = 0; MvD cm;
    if (temporal_mv(e, r, 0, 0, &cm)) { dir |= 1; ml->f[cnt][0].mv = cm; ml->f[cnt][0].ref = 0; }
    if (isB && temporal_mv(e, r, 1, 0, &cm)) { dir |= 2; ml->f[cnt][1].mv = cm; ml->f[cnt][1].ref = 0; }
    if (dir) { ml->dir[cnt] = (uint8_t)dir; cnt++; }
  }
  if (cnt == maxC) return;
  int arr = cnt; const int cutoff = arr;
  if (isB) { // combined bi-predictive candidates :2556-2597 (every candidate so far is an inter candidate)
    for (int idx = 0; idx < cutoff * (cutoff - 1) && arr != maxC; idx++) {
      // uiPriorityList0 = {0,1,0,2,1,2,0,3,1,3,2,3}, uiPriorityList1 = {1,0,2,0,2,1,3,0,3,1,3,2}
      const int pi = idx == 0 ? 0 : idx == 1 ? 1 : idx == 2 ? 0 : idx == 3 ? 2 : idx == 4 ? 1 : idx == 5 ? 2 : idx == 6 ? 0 : idx == 7 ? 3 : idx == 8 ? 1 : idx == 9 ? 3 : idx == 10 ? 2 : 3;
      const int pj = idx == 0 ? 1 : idx == 1 ? 0 : idx == 2 ? 2 : idx == 3 ? 0 : idx == 4 ? 2 : idx == 5 ? 1 : idx == 6 ? 3 : idx == 7 ? 0 : idx == 8 ? 3 : idx == 9 ? 1 : idx == 10 ? 3 : 2;
      if ((ml->dir[pi] & 1) && (ml->dir[pj] & 2)) {
        ml->dir[arr] = 3; ml->f[arr][0] = ml->f[pi][0]; ml->f[arr][1] = ml->f[pj][1];
        const int p0 = s->ref[0][ml->f[arr][0].ref].poc, p1 = s->ref[1][ml->f[arr][1].ref].poc;
        if (!(p0 == p1 && ml->f[arr][0].mv.x == ml->f[arr][1].mv.x && ml->f[arr][0].mv.y == ml->f[arr][1].mv.y)) arr++;
      }
    }
  }
  if (arr == maxC) return;
  int rr = 0, refcnt = 0;
  const int numRef = isB ? (s->numRefIdx[0] < s->numRefIdx[1] ? s->numRefIdx[0] : s->numRefIdx[1]) : s->numRefIdx[0];
  while (arr < maxC) {
    ml->dir[arr] = 1; ml->f[arr][0].mv.x = ml->f[arr][0].mv.y = 0; ml->f[arr][0].ref = rr;
    if (isB) { ml->dir[arr] = 3; ml->f[arr][1].mv.x = ml->f[arr][1].mv.y = 0; ml->f[arr][1].ref = rr; }
    arr++;
    if (refcnt == numRef - 1) rr = 0; else { ++rr; ++refcnt; }
  }
  ml->num = arr;
  HM_TRACE(e, 5, ((uint32_t)(uint16_t)ml->f[0][0].mv.x << 16) | (uint16_t)ml->f[0][0].mv.y, ((uint32_t)(uint16_t)ml->f[1][0].mv.x << 16) | (uint16_t)ml->f[1][0].mv.y, (double)arr);
}

// ------------------------------------------------------------------------------------------------
// AMVP candidates (fillMvpCand :2752, xAddMVPCand :2978, xAddMVPCandOrder :3064)
// ------------------------------------------------------------------------------------------------
HM_DEV inline int add_mvp_cand(const Shared *e, AmvpInfo *info, int list, int refIdx, int x4, int y4, int curZ, int dir)
{
  const InterPic *s = e->fb.ip; int z;
  const int ca = nb_at(e, x4, y4, dir, curZ, &z);
  if (ca < 0) return 0;
  const InterMeta *m = imeta_of(e, ca);
  const int curRefPOC = s->ref[list][refIdx].poc;
  if (m->refIdx[list][z] >= 0 && curRefPOC == s->ref[list][m->refIdx[list][z]].poc) { info->cand[info->n++] = m->mv[list][z]; return 1; }
  const int l2 = 1 - list;
  if (m->refIdx[l2][z] >= 0 && s->ref[l2][m->refIdx[l2][z]].poc == curRefPOC) { info->cand[info->n++] = m->mv[l2][z]; return 1; }
  return 0;
}
HM_DEV inline int add_mvp_cand_order(const Shared *e, AmvpInfo *info, int list, int refIdx, int x4, int y4, int curZ, int dir)
{
  const InterPic *s = e->fb.ip; int z;
  const int ca = nb_at(e, x4, y4, dir, curZ, &z);
  if (ca < 0) return 0;
  const InterMeta *m = imeta_of(e, ca);
  const int curRefPOC = s->ref[list][refIdx].poc, curLT = s->ref[list][refIdx].isLongTerm;
  for (int k = 0; k < 2; k++) {
    const int l = k ? 1 - list : list;
    if (m->refIdx[l][z] >= 0) {
      const RefPicDev *nr = &s->ref[l][m->refIdx[l][z]];
      if (curLT == nr->isLongTerm) {
        MvD mv = m->mv[l][z];
        if (!curLT) { const int scale = dist_scale_factor(s->poc, curRefPOC, s->poc, nr->poc); if (scale != 4096) mv = scale_mv(mv, scale); }
        info->cand[info->n++] = mv;
        return 1;
      }
    }
  }
  return 0;
}
HM_DEV HM_NOINLINE void fill_mvp_cand(Shared *e, int cuZ, int cuDepth, int partSize, int puIdx, int list, int refIdx, AmvpInfo *info)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); puIdx = HM_UNI(puIdx); list = HM_UNI(list); refIdx = HM_UNI(refIdx); info = hm_uni_ptr(info);
  const InterPic *s = e->fb.ip;
  info->n = 0;
  if (refIdx < 0) return;
  const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
  const int ltx = r.x >> 2, lty = r.y >> 2, rtx = (r.x + r.w - 4) >> 2, lbx = ltx, lby = (r.y + r.h - 4) >> 2;
  const int zLT = hm_r2z(lty * 16 + ltx), zRT = hm_r2z(lty * 16 + rtx), zLB = hm_r2z(lby * 16 + lbx);
  int z, added, addedSmvp;
  int ca = nb_at(e, lbx, lby, NB_BELOW_LEFT, zLB, &z);
  addedSmvp = ca >= 0 && nb_is_inter(e, ca, z);
  if (!addedSmvp) { ca = nb_at(e, lbx, lby, NB_LEFT, zLB, &z); addedSmvp = ca >= 0 && nb_is_inter(e, ca, z); }
  added = add_mvp_cand(e, info, list, refIdx, lbx, lby, zLB, NB_BELOW_LEFT);
  if (!added) added = add_mvp_cand(e, info, list, refIdx, lbx, lby, zLB, NB_LEFT);
  if (!added) { added = add_mvp_cand_order(e, info, list, refIdx, lbx, lby, zLB, NB_BELOW_LEFT); if (!added) add_mvp_cand_order(e, info, list, refIdx, lbx, lby, zLB, NB_LEFT); }
  added = add_mvp_cand(e, info, list, refIdx, rtx, lty, zRT, NB_ABOVE_RIGHT);
  if (!added) added = add_mvp_cand(e, info, list, refIdx, rtx, lty, zRT, NB_ABOVE);
  if (!added) add_mvp_cand(e, info, list, refIdx, ltx, lty, zLT, NB_ABOVE_LEFT);
  if (!addedSmvp) {
    added = add_mvp_cand_order(e, info, list, refIdx, rtx, lty, zRT, NB_ABOVE_RIGHT);
    if (!added) added = add_mvp_cand_order(e, info, list, refIdx, rtx, lty, zRT, NB_ABOVE);
    if (!added) add_mvp_cand_order(e, info, list, refIdx, ltx, lty, zLT, NB_ABOVE_LEFT);
  }
  if (info->n == 2 && info->cand[0].x == info->cand[1].x && info->cand[0].y == info->cand[1].y) info->n = 1;
  if (s->tmvp) { MvD cm; if (temporal_mv(e, r, list, refIdx, &cm)) info->cand[info->n++] = cm; }
  if (info->n > 2) info->n = 2;
  while (info->n < 2) { info->cand[info->n].x = info->cand[info->n].y = 0; info->n++; }
  HM_TRACE(e, 4, ((uint32_t)(uint16_t)info->cand[0].x << 16) | (uint16_t)info->cand[0].y, ((uint32_t)(uint16_t)info->cand[1].x << 16) | (uint16_t)info->cand[1].y, (double)refIdx);
}

// ------------------------------------------------------------------------------------------------
// motion cost (TComRdCost.h:160-189)
// ------------------------------------------------------------------------------------------------
HM_DEV inline uint32_t mv_comp_bits(int val)
{
  const uint32_t tmp = (val <= 0) ? (uint32_t)((-val << 1) + 1) : (uint32_t)(val << 1);
  return 1u + 2u * (uint32_t)(31 - __builtin_clz(tmp));          // xGetComponentBits (TComRdCost.cpp:278): one shift per 2 bits until tmp == 1
}
HM_DEV inline uint32_t mc_bits(const Shared *e, int x, int y)
{ return mv_comp_bits((x << e->costScale) - e->mvPredictor.x) + mv_comp_bits((y << e->costScale) - e->mvPredictor.y); }
HM_DEV inline uint32_t mc_cost32(const Shared *e, uint32_t b) { return (uint32_t)(e->mcost * b) >> 16; }

// ------------------------------------------------------------------------------------------------
// integer search: TZ (xTZSearch :4027-4228, helpers :333-795)
// ------------------------------------------------------------------------------------------------
// Search points (xTZSearchHelp :333) are queued and evaluated in batches: the points of one diamond / 2-point / raster pattern do not
// depend on each other's result, only the running best does.  One pass takes up to 16 queued points: the 64 lanes are split into as many
// groups as there are points (4..64 lanes per point), a group accumulates its point's SAD over the (row sub-sampled) block with coalesced
// row reads, a DPP prefix inside the group leaves the sum in its last lane, which also prices the motion vector; the pass's winner is the
// cheapest point, the first one on a tie -- what the reference's one-by-one update with strict "<" arrives at -- and the running best is
// updated once per pass on the scalar unit.
// LDS-staged reference window (north_star: "LDS-staged pixel tiles" for block matching): the samples every search point within R of a centre can
// touch -- (w + 2R) x (h + 2R), clipped to the search range -- are copied once into the transform buffers (idle during the motion search: 8.4 KB),
// eight loads in flight per lane, and the passes whose points all lie inside read LDS instead of the reference plane.  Used where every point is
// known to fall inside and the points are many: the +-4 full search of the bi-prediction refinement (xPatternSearch :3932, 81 points from one
// (w + 8) x (h + 8) window).  Measured and NOT used for the TZ diamonds (xTZSearch :4027): their 4-8 points per round read L2-resident rows with the
// loads of a pass in flight together, and staging a +-16 window cost more than the rounds it served saved (45 k instead of 29 k cycles per search);
// the raster scan would need (w + 128) x (h + 128).
HM_DEV inline void tz_stage_window(Shared *e, TZ *z, int cx, int cy, int R)
{
  const int w = z->w, h = z->h;
  z->ww = 0;
  if ((w + 2 * R) * (h + 2 * R) > HM_TZ_WIN_SAMPLES) return;
  const int x0 = cx - R > z->l ? cx - R : z->l, x1 = cx + R < z->r ? cx + R : z->r, y0 = cy - R > z->t ? cy - R : z->t, y1 = cy + R < z->b ? cy + R : z->b;
  if (x1 < x0 || y1 < y0) return;
  const int ww = x1 - x0 + w, wh = y1 - y0 + h, sr = z->refStride, total = ww * wh;
  const Pel *src = z->ref + (ptrdiff_t)y0 * sr + x0; Pel *win = (Pel *)e->bufA;
  for (int base = 0; base < total; base += 8 * HM_NT) {
    Pel v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int i = base + k * HM_NT + hm_lane(); const int y = i / ww, x = i - y * ww; v[k] = i < total ? src[(ptrdiff_t)y * sr + x] : (Pel)0; }
#pragma unroll
    for (int k = 0; k < 8; k++) { const int i = base + k * HM_NT + hm_lane(); if (i < total) win[i] = v[k]; }
  }
  z->wx0 = (int16_t)x0; z->wy0 = (int16_t)y0; z->ww = (int16_t)ww; z->wh = (int16_t)wh;
  HM_SYNC();
}
#define TZ_PUSH(z, n, X, Y, PN, D) do { (z)->lx[n] = (int16_t)(X); (z)->ly[n] = (int16_t)(Y); (z)->lp[n] = (int8_t)(PN); (z)->ld[n] = (int8_t)(D); (n)++; } while (0)
HM_DEV HM_NOINLINE void tz_eval_list(Shared *e, int n)
{
  HM_ENTRY(e); n = HM_UNI(n);
  TZ *z = &e->tz;
  const Pel *org = hm_uni_ptr(z->org), *ref = hm_uni_ptr(z->ref);
  const int so = HM_UNI(z->orgStride), sr = HM_UNI(z->refStride), w = HM_UNI(z->w), h = HM_UNI(z->h), sub = HM_UNI(z->subShift), bd = e->bitDepth;
  const int rows = h >> sub, so2 = so << sub, sr2 = sr << sub, S = rows * w;
  const int groups = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : (n <= 8 ? 8 : 16))), L = 64 / groups;      // lanes per point
  // all points of the pass inside the staged window?
  const int ww = HM_UNI(z->ww), wx0 = HM_UNI(z->wx0), wy0 = HM_UNI(z->wy0), wh = HM_UNI(z->wh);
  int inWin = ww != 0;
  for (int pnt = 0; pnt < n && inWin; pnt++) { const int px = z->lx[pnt], py = z->ly[pnt]; inWin = px >= wx0 && px <= wx0 + ww - w && py >= wy0 && py <= wy0 + wh - h; }
  inWin = HM_UNI(inWin);
  HM_LV(int32_t, vCost);
  if (inWin) {
    const Pel *win = (const Pel *)e->bufA; const int sw2 = ww << sub;
    HM_WAVE_FOR(k) {
      const int pnt = k / L, j = k - pnt * L;
      uint32_t sum = 0;
      if (pnt < n) {
        const Pel *r = win + (z->ly[pnt] - wy0) * ww + (z->lx[pnt] - wx0);
        const int sy = L / w, sx = L - sy * w;
        int yy = j / w, x = j - yy * w;
        for (int i = j; i < S; i += L) {
          sum += (uint32_t)hm_abs(org[yy * so2 + x] - r[yy * sw2 + x]);
          x += sx; yy += sy; if (x >= w) { x -= w; yy++; }
        }
      }
      HM_LVK(vCost, k) = (int32_t)sum;
    }
  } else {
    HM_WAVE_FOR(k) {
      const int pnt = k / L, j = k - pnt * L;
      uint32_t sum = 0;
      if (pnt < n) {
        const Pel *r = ref + (ptrdiff_t)z->ly[pnt] * sr + z->lx[pnt];
        const int sy = L / w, sx = L - sy * w;
        int yy = j / w, x = j - yy * w;
        for (int i = j; i < S; i += L) {
          sum += (uint32_t)hm_abs(org[yy * so2 + x] - r[yy * sr2 + x]);
          x += sx; yy += sy; if (x >= w) { x -= w; yy++; }
        }
      }
      HM_LVK(vCost, k) = (int32_t)sum;
    }
  }
#ifdef HM355_HOSTSIM
  for (int pnt = 0; pnt < groups; pnt++) { int32_t t = 0; for (int j = 0; j < L; j++) t += vCost[pnt * L + j]; vCost[pnt * L + L - 1] = t; }
#else
  { // inclusive prefix over the lanes of a group: its last lane ends up with the group's sum (row_shr never reaches past a 16-lane row,
    // groups of 32 / 64 lanes add the row sums)
    int v = vCost;
    if (L >= 2) v += hm_dpp<0x111>(0, v);
    if (L >= 4) v += hm_dpp<0x112>(0, v);
    if (L >= 8) v += hm_dpp<0x114>(0, v);
    if (L >= 16) v += hm_dpp<0x118>(0, v);
    if (L == 32) { const int a = __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31), b = __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63); v = hm_lane() < 32 ? a : b; }
    if (L == 64) v = __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63);
    vCost = v;
  }
#endif
  HM_WAVE_FOR(k) { // the last lane of a group: SAD -> cost of its point (TComRdCost::getCost(x, y), TComRdCost.h:160-189)
    const int pnt = k / L;
    int32_t c = 0x7fffffff;
    if (pnt < n && k == pnt * L + L - 1) {
      const uint32_t sad = ((uint32_t)HM_LVK(vCost, k) << sub) >> (bd - 8);
      HM_TRACE(e, 11, ((uint32_t)(uint16_t)z->lx[pnt] << 16) | (uint16_t)z->ly[pnt], sad, (double)z->bestSad);
      c = (int32_t)(sad + mc_cost32(e, mc_bits(e, z->lx[pnt], z->ly[pnt])));
    }
    HM_LVK(vCost, k) = c;
  }
  int best, bestPnt;
#ifdef HM355_HOSTSIM
  best = 0x7fffffff; bestPnt = 0;
  for (int pnt = 0; pnt < n; pnt++) if (vCost[pnt * L + L - 1] < best) { best = vCost[pnt * L + L - 1]; bestPnt = pnt; }
#else
  best = 0x7fffffff - hm_wave_max_i(0x7fffffff - vCost);
  bestPnt = (int)(__builtin_ctzll(__ballot(vCost == best))) / L;
#endif
  if ((uint32_t)best < z->bestSad) { z->bestSad = (uint32_t)best; z->bestX = z->lx[bestPnt]; z->bestY = z->ly[bestPnt]; z->bestDist = z->ld[bestPnt]; z->bestRound = 0; z->pointNr = z->lp[bestPnt]; }
  HM_SYNC();
}
HM_DEV inline void tz_help(Shared *e, int sx, int sy, int pointNr, int dist)
{ TZ *z = &e->tz; int n = 0; TZ_PUSH(z, n, sx, sy, pointNr, dist); tz_eval_list(e, n); }
HM_DEV inline void tz_2point(Shared *e, TZ *z)
{
  const int x = z->bestX, y = z->bestY; int n_ = 0;
  switch (z->pointNr) {
    case 1: if (x - 1 >= z->l) TZ_PUSH(z, n_, x - 1, y, 0, 2); if (y - 1 >= z->t) TZ_PUSH(z, n_, x, y - 1, 0, 2); break;
    case 2: if (y - 1 >= z->t) { if (x - 1 >= z->l) TZ_PUSH(z, n_, x - 1, y - 1, 0, 2); if (x + 1 <= z->r) TZ_PUSH(z, n_, x + 1, y - 1, 0, 2); } break;
    case 3: if (y - 1 >= z->t) TZ_PUSH(z, n_, x, y - 1, 0, 2); if (x + 1 <= z->r) TZ_PUSH(z, n_, x + 1, y, 0, 2); break;
    case 4: if (x - 1 >= z->l) { if (y + 1 <= z->b) TZ_PUSH(z, n_, x - 1, y + 1, 0, 2); if (y - 1 >= z->t) TZ_PUSH(z, n_, x - 1, y - 1, 0, 2); } break;
    case 5: if (x + 1 <= z->r) { if (y - 1 >= z->t) TZ_PUSH(z, n_, x + 1, y - 1, 0, 2); if (y + 1 <= z->b) TZ_PUSH(z, n_, x + 1, y + 1, 0, 2); } break;
    case 6: if (x - 1 >= z->l) TZ_PUSH(z, n_, x - 1, y, 0, 2); if (y + 1 <= z->b) TZ_PUSH(z, n_, x, y + 1, 0, 2); break;
    case 7: if (y + 1 <= z->b) { if (x - 1 >= z->l) TZ_PUSH(z, n_, x - 1, y + 1, 0, 2); if (x + 1 <= z->r) TZ_PUSH(z, n_, x + 1, y + 1, 0, 2); } break;
    case 8: if (x + 1 <= z->r) TZ_PUSH(z, n_, x + 1, y, 0, 2); if (y + 1 <= z->b) TZ_PUSH(z, n_, x, y + 1, 0, 2); break;
    default: break;
  }
  if (n_) tz_eval_list(e, n_);
}
HM_DEV inline void tz_diamond(Shared *e, TZ *z, int sx, int sy, int d)
{
  const int top = sy - d, bot = sy + d, lef = sx - d, rig = sx + d;
  z->bestRound += 1;
  int n_ = 0;
  const int zl_ = HM_UNI(z->l), zr_ = HM_UNI(z->r), zt_ = HM_UNI(z->t), zb_ = HM_UNI(z->b);
  if (d == 1) {
    if (top >= zt_) TZ_PUSH(z, n_, sx, top, 2, d);
    if (lef >= zl_) TZ_PUSH(z, n_, lef, sy, 4, d);
    if (rig <= zr_) TZ_PUSH(z, n_, rig, sy, 5, d);
    if (bot <= zb_) TZ_PUSH(z, n_, sx, bot, 7, d);
  } else if (d <= 8) {
    const int top2 = sy - (d >> 1), bot2 = sy + (d >> 1), lef2 = sx - (d >> 1), rig2 = sx + (d >> 1);
    if (top >= zt_ && lef >= zl_ && rig <= zr_ && bot <= zb_) {
      TZ_PUSH(z, n_, sx, top, 2, d); TZ_PUSH(z, n_, lef2, top2, 1, d >> 1); TZ_PUSH(z, n_, rig2, top2, 3, d >> 1); TZ_PUSH(z, n_, lef, sy, 4, d);
      TZ_PUSH(z, n_, rig, sy, 5, d); TZ_PUSH(z, n_, lef2, bot2, 6, d >> 1); TZ_PUSH(z, n_, rig2, bot2, 8, d >> 1); TZ_PUSH(z, n_, sx, bot, 7, d);
    } else {
      if (top >= zt_) TZ_PUSH(z, n_, sx, top, 2, d);
      if (top2 >= zt_) { if (lef2 >= zl_) TZ_PUSH(z, n_, lef2, top2, 1, d >> 1); if (rig2 <= zr_) TZ_PUSH(z, n_, rig2, top2, 3, d >> 1); }
      if (lef >= zl_) TZ_PUSH(z, n_, lef, sy, 4, d);
      if (rig <= zr_) TZ_PUSH(z, n_, rig, sy, 5, d);
      if (bot2 <= zb_) { if (lef2 >= zl_) TZ_PUSH(z, n_, lef2, bot2, 6, d >> 1); if (rig2 <= zr_) TZ_PUSH(z, n_, rig2, bot2, 8, d >> 1); }
      if (bot <= zb_) TZ_PUSH(z, n_, sx, bot, 7, d);
    }
  } else {
    if (top >= zt_ && lef >= zl_ && rig <= zr_ && bot <= zb_) {
      TZ_PUSH(z, n_, sx, top, 0, d); TZ_PUSH(z, n_, lef, sy, 0, d); TZ_PUSH(z, n_, rig, sy, 0, d); TZ_PUSH(z, n_, sx, bot, 0, d);
      for (int i = 1; i < 4; i++) {
        const int yt = top + ((d >> 2) * i), yb = bot - ((d >> 2) * i), xl = sx - ((d >> 2) * i), xr = sx + ((d >> 2) * i);
        TZ_PUSH(z, n_, xl, yt, 0, d); TZ_PUSH(z, n_, xr, yt, 0, d); TZ_PUSH(z, n_, xl, yb, 0, d); TZ_PUSH(z, n_, xr, yb, 0, d);
      }
    } else {
      if (top >= zt_) TZ_PUSH(z, n_, sx, top, 0, d);
      if (lef >= zl_) TZ_PUSH(z, n_, lef, sy, 0, d);
      if (rig <= zr_) TZ_PUSH(z, n_, rig, sy, 0, d);
      if (bot <= zb_) TZ_PUSH(z, n_, sx, bot, 0, d);
      for (int i = 1; i < 4; i++) {
        const int yt = top + ((d >> 2) * i), yb = bot - ((d >> 2) * i), xl = sx - ((d >> 2) * i), xr = sx + ((d >> 2) * i);
        if (yt >= zt_) { if (xl >= zl_) TZ_PUSH(z, n_, xl, yt, 0, d); if (xr <= zr_) TZ_PUSH(z, n_, xr, yt, 0, d); }
        if (yb <= zb_) { if (xl >= zl_) TZ_PUSH(z, n_, xl, yb, 0, d); if (xr <= zr_) TZ_PUSH(z, n_, xr, yb, 0, d); }
      }
    }
  }
  if (n_) tz_eval_list(e, n_);
}
HM_DEV inline void set_search_range(const Shared *e, MvD pred, int rng, int cuX, int cuY, MvD *lt, MvD *rb)
{ // xSetSearchRange :3911
  const MvD p = clip_mv(e, pred, cuX, cuY);
  MvD a, b; a.x = (int16_t)(p.x - (rng << 2)); a.y = (int16_t)(p.y - (rng << 2)); b.x = (int16_t)(p.x + (rng << 2)); b.y = (int16_t)(p.y + (rng << 2));
  a = clip_mv(e, a, cuX, cuY); b = clip_mv(e, b, cuX, cuY);
  lt->x = a.x >> 2; lt->y = a.y >> 2; rb->x = b.x >> 2; rb->y = b.y >> 2;
}
HM_DEV inline uint32_t tz_search(Shared *e, TZ *z, MvD *mv, int cuX, int cuY, MvD lt, MvD rb, int useInt, MvD intMv2Nx2N)
{
  const int searchRange = 64, raster = 5;
  int rl = lt.x, rr = rb.x, rt = lt.y, rbm = rb.y;
  z->l = lt.x; z->r = rb.x; z->t = lt.y; z->b = rb.y;
  MvD c = clip_mv(e, *mv, cuX, cuY); c.x >>= 2; c.y >>= 2;
  z->bestSad = 0xffffffffu; z->bestX = z->bestY = 0; z->bestDist = 0; z->bestRound = 0; z->pointNr = 0; z->ww = 0;
  { // the start points (:4066-4090): predictor, zero vector, the 2Nx2N integer vector -- independent evaluations, one pass
    int n_ = 0;
    TZ_PUSH(z, n_, c.x, c.y, 0, 0); TZ_PUSH(z, n_, 0, 0, 0, 0);
    if (useInt) {
      MvD im; im.x = (int16_t)(intMv2Nx2N.x << 2); im.y = (int16_t)(intMv2Nx2N.y << 2);
      im = clip_mv(e, im, cuX, cuY); im.x >>= 2; im.y >>= 2;
      TZ_PUSH(z, n_, im.x, im.y, 0, 0);
    }
    tz_eval_list(e, n_);
  }
  if (useInt) {
    MvD nb, nlt, nrb; nb.x = (int16_t)(z->bestX << 2); nb.y = (int16_t)(z->bestY << 2);
    set_search_range(e, nb, searchRange, cuX, cuY, &nlt, &nrb);
    rl = nlt.x; rr = nrb.x; rt = nlt.y; rbm = nrb.y;
  }
  int sx = z->bestX, sy = z->bestY, d;
  for (d = 1; d <= searchRange; d *= 2) { tz_diamond(e, z, sx, sy, d); if (z->bestRound >= 3) break; }
  if (z->bestDist == 1) { z->bestDist = 0; tz_2point(e, z); }
  if (z->bestDist > raster) {
    z->bestDist = raster;
    int n_ = 0;
    for (sy = rt; sy <= rbm; sy += raster) for (sx = rl; sx <= rr; sx += raster) { TZ_PUSH(z, n_, sx, sy, 0, raster); if (n_ == 16) { tz_eval_list(e, n_); n_ = 0; } }
    if (n_) tz_eval_list(e, n_);
  }
  while (z->bestDist > 0) {
    sx = z->bestX; sy = z->bestY;
    z->bestDist = 0; z->pointNr = 0;
    for (d = 1; d < searchRange + 1; d *= 2) tz_diamond(e, z, sx, sy, d);
    if (z->bestDist == 1) { z->bestDist = 0; if (z->pointNr != 0) tz_2point(e, z); }
  }
  mv->x = (int16_t)z->bestX; mv->y = (int16_t)z->bestY;
  HM_TRACE(e, 10, ((uint32_t)(uint16_t)z->bestX << 16) | (uint16_t)z->bestY, z->bestSad, 0.0);
  return z->bestSad - mc_cost32(e, mc_bits(e, z->bestX, z->bestY));
}

// ------------------------------------------------------------------------------------------------
// fractional refinement (xPatternSearchFracDIF :4386, xPatternRefinement :799): the candidate block at a quarter-sample
// displacement is the two-stage interpolation of the reference, compared with SATD (HadamardME)
// ------------------------------------------------------------------------------------------------
HM_CONST int8_t HM_MV_REFINE_H[9][2] = { {0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1} };
HM_CONST int8_t HM_MV_REFINE_Q[9][2] = { {0, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {1, 1} };
HM_DEV inline uint32_t pattern_refinement(Shared *e, TZ *z, const Pel *refAtInt, MvD base, int frac, MvD *mvFrac)
{
  // The nine candidates are (tx, ty) in {-1,0,1}^2 around `base`.  Candidates with the same tx share the first (horizontal)
  // interpolation stage, so it is done once per tx over the rows all three ty need (two integer row offsets at most);
  // the costs are then compared in the reference's candidate order (strict <, first wins).
  const int bd = e->bitDepth, w = z->w, h = z->h, rs = z->refStride;
  const int vyLo = (base.y - 1) * frac, rowLo = vyLo >> 2, rows = h + 8;      // rows rowLo-3 .. rowLo+h+4 of the reference
  // the two interpolation buffers live in LDS (the transform buffers are idle during the motion search) whenever the block fits:
  // every PU up to 32x32 and the 16-wide AMP parts; only the 48- and 64-wide PUs go through the HBM workspace
  const int inLds = w * rows <= (int)(sizeof(e->bufA) / sizeof(Pel)) && w * h <= (int)(sizeof(e->u.bufB) / sizeof(Pel));
  Pel *tmp = inLds ? (Pel *)e->bufA : e->ws->mcTmp, *blk = inLds ? (Pel *)e->u.bufB : e->ws->mcBlk;
  const int ts = inLds ? w : 64;                                               // row stride of both buffers
  int32_t *cost9 = e->absCoeff;                                                // [(ty+1)*3 + tx+1]
  for (int tx = -1; tx <= 1; tx++) {
    const int hx = (tx + base.x) * frac, xFrac = hx & 3;
    const Pel *r = refAtInt + (ptrdiff_t)(rowLo - 3) * rs + (hx >> 2);
    HM_PAR_FOR_XY(x, y, w, w * rows) tmp[y * ts + x] = if_sample(bd, r + (ptrdiff_t)y * rs + x, 1, xFrac, 0, 1, 0);
    HM_SYNC();
    for (int ty = -1; ty <= 1; ty++) {
      const int vy = (ty + base.y) * frac, yFrac = vy & 3, ro = (vy >> 2) - rowLo;
      HM_PAR_FOR_XY(x, y, w, w * h) blk[y * ts + x] = if_sample(bd, tmp + (y + ro + 3) * ts + x, ts, yFrac, 0, 0, 1);
      HM_SYNC();
      uint32_t d = dist_hads_rect(z->org, z->orgStride, blk, ts, w, h, bd);
      d += mc_cost32(e, mc_bits(e, tx + mvFrac->x, ty + mvFrac->y));
      cost9[(ty + 1) * 3 + tx + 1] = (int32_t)d;
    }
  }
  HM_SYNC();
  uint32_t best = 0xffffffffu; int bestDir = 0;
  for (int i = 0; i < 9; i++) {
    const int tx = frac == 2 ? HM_MV_REFINE_H[i][0] : HM_MV_REFINE_Q[i][0], ty = frac == 2 ? HM_MV_REFINE_H[i][1] : HM_MV_REFINE_Q[i][1];
    const uint32_t d = (uint32_t)cost9[(ty + 1) * 3 + tx + 1];
    if (d < best) { best = d; bestDir = i; }
  }
  mvFrac->x = frac == 2 ? HM_MV_REFINE_H[bestDir][0] : HM_MV_REFINE_Q[bestDir][0];
  mvFrac->y = frac == 2 ? HM_MV_REFINE_H[bestDir][1] : HM_MV_REFINE_Q[bestDir][1];
  return best;
}

// xPatternSearch :3932-3988: full search over the (small) bi-prediction range, one point per tz_help call
HM_DEV inline uint32_t pattern_search(Shared *e, TZ *z, MvD *mv, MvD lt, MvD rb)
{
  z->bestSad = 0xffffffffu; z->bestX = z->bestY = 0; z->bestDist = 0; z->bestRound = 0; z->pointNr = 0; z->ww = 0;
  z->l = lt.x; z->r = rb.x; z->t = lt.y; z->b = rb.y;
  tz_stage_window(e, z, (lt.x + rb.x) >> 1, (lt.y + rb.y) >> 1, 5);     // all 81 points from one window
  int n_ = 0;
  for (int y = lt.y; y <= rb.y; y++) for (int x = lt.x; x <= rb.x; x++) { TZ_PUSH(z, n_, x, y, 0, 0); if (n_ == 16) { tz_eval_list(e, n_); n_ = 0; } }
  if (n_) tz_eval_list(e, n_);
  mv->x = (int16_t)z->bestX; mv->y = (int16_t)z->bestY;
  return z->bestSad - mc_cost32(e, mc_bits(e, z->bestX, z->bestY));
}
// xMotionEstimation :3816-3906; results in e->outMv / e->outBits / e->outDist(cost).
// bi != 0: (inX, inY) is this list's uni-directional MV (search centre, quarter samples) and the pattern is
// 2*org - prediction of the other list (ws->yuvPred[1-list]), TComYuv::removeHighFreq :393 without clipping
HM_DEV HM_NOINLINE void motion_estimation(Shared *e, int cuZ, int cuDepth, int partSize, int puIdx, int predX, int predY, int listRef, uint32_t bitsIn, int bi, int inX, int inY, int intOnly = 0)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); puIdx = HM_UNI(puIdx); predX = HM_UNI(predX); predY = HM_UNI(predY); listRef = HM_UNI(listRef); bitsIn = HM_UCALL(bitsIn);
  bi = HM_UNI(bi); inX = HM_UNI(inX); inY = HM_UNI(inY); intOnly = HM_UNI(intOnly);
  const int list = listRef >> 4, refIdx = listRef & 15;
  InterPic *s = e->fb.ip;
  const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
  const int rz = hm_z2r(cuZ), cuX = e->ctuX * 64 + (rz & 15) * 4, cuY = e->ctuY * 64 + (rz >> 4) * 4;
  const int px = e->ctuX * 64 + r.x, py = e->ctuY * 64 + r.y;
  const RefPicDev *ref = &s->ref[list][refIdx];
  MvD mvPred; mvPred.x = (int16_t)predX; mvPred.y = (int16_t)predY;
  TZ &z = e->tz;
  z.org = e->fb.org[0] + (ptrdiff_t)py * e->stride[0] + px; z.orgStride = e->stride[0]; z.w = r.w; z.h = r.h;
  if (bi) {
    const Pel *other = e->ws->yuvPred[1 - list], *org = z.org; Pel *ob = e->ws->orgBi; const int so = z.orgStride;
    HM_PAR_FOR_XY(x, y, r.w, r.w * r.h) ob[(r.y + y) * 64 + r.x + x] = (Pel)(2 * org[y * so + x] - other[(r.y + y) * 64 + r.x + x]);
    HM_SYNC();
    z.org = ob + r.y * 64 + r.x; z.orgStride = 64;
  }
  z.ref = ref->plane[0] + (ptrdiff_t)py * ref->stride[0] + px; z.refStride = ref->stride[0];
  z.subShift = r.h > 8 ? 1 : 0;
  MvD lt, rb, centre = mvPred;
  if (bi) { centre.x = (int16_t)inX; centre.y = (int16_t)inY; }
  set_search_range(e, centre, bi ? 4 : 64, cuX, cuY, &lt, &rb);   // BipredSearchRange 4 / SearchRange 64
  e->mcost = s->lambdaMotionSAD; e->mvPredictor = mvPred; e->costScale = 2;
  MvD mv = mvPred;
  uint32_t c;
  HM_PROF_BEGIN(e, PR_ME_INT);
  if (bi) c = pattern_search(e, &z, &mv, lt, rb);
  else {
    const int useInt = (partSize != SIZE_2Nx2N || cuDepth != 0);
    MvD im = e->ws->intMv[list][refIdx];
    c = tz_search(e, &z, &mv, cuX, cuY, lt, rb, useInt, im);
    if (partSize == SIZE_2Nx2N) e->ws->intMv[list][refIdx] = mv;
  }
  HM_PROF_END(e, PR_ME_INT);
  if (intOnly) return;                       // me_token_prepass: only m_integerMv2Nx2N is wanted
  HM_PROF_BEGIN(e, PR_ME_FRAC);
  e->costScale = 1;
  const Pel *refAtInt = z.ref + (ptrdiff_t)mv.y * z.refStride + mv.x;
  MvD half, base, qter; half.x = (int16_t)(mv.x << 1); half.y = (int16_t)(mv.y << 1); base.x = base.y = 0;
  c = pattern_refinement(e, &z, refAtInt, base, 2, &half);
  e->costScale = 0;
  base.x = (int16_t)(half.x << 1); base.y = (int16_t)(half.y << 1);
  qter.x = (int16_t)(((mv.x << 1) + half.x) << 1); qter.y = (int16_t)(((mv.y << 1) + half.y) << 1);
  c = pattern_refinement(e, &z, refAtInt, base, 1, &qter);
  mv.x = (int16_t)((mv.x << 2) + (half.x << 1) + qter.x); mv.y = (int16_t)((mv.y << 2) + (half.y << 1) + qter.y);
  HM_PROF_END(e, PR_ME_FRAC);
  const uint32_t mvBits = mc_bits(e, mv.x, mv.y);
  const uint32_t bits = bitsIn + mvBits;
  e->outMv = mv; e->outBits = bits;
  e->outDist = (uint32_t)(floor((bi ? 0.5 : 1.0) * ((double)c - (double)mc_cost32(e, mvBits))) + (double)mc_cost32(e, bits));
  HM_TRACE(e, 2, ((uint32_t)(uint16_t)mv.x << 16) | (uint16_t)mv.y, bits, (double)e->outDist);
}

// xPredInterUni :586: one list of one PU into a CTU-relative scratch picture (bi: 14-bit intermediate, no rounding)
HM_DEV inline void pred_inter_uni(Shared *e, int cuZ, Rect r, int list, Pel *dst, int bi)
{
  const InterPic *s = e->fb.ip; const InterMeta *m = e->im;
  const int z = rect_z(r), rz = hm_z2r(cuZ), cuX = e->ctuX * 64 + (rz & 15) * 4, cuY = e->ctuY * 64 + (rz >> 4) * 4;
  const int refIdx = m->refIdx[list][z];
  const MvD mv = clip_mv(e, m->mv[list][z], cuX, cuY);
  for (int c = 0; c < 3; c++) {
    const int st = HM_PLANE_STRIDE(c), sh = c ? 1 : 0;
    pred_inter_blk(e, c, &s->ref[list][refIdx], e->ctuX * 64 + r.x, e->ctuY * 64 + r.y, mv, r.w, r.h, dst + HM_PLANE_OFF(c) + (r.y >> sh) * st + (r.x >> sh), st, bi);
  }
}
// TComPrediction::motionCompensation :514 of one PU into a CTU-relative scratch picture.  list 0 / 1: that list alone
// (uni-directional, rounded); list 2 = REF_PIC_LIST_X: what the PU's motion says (xCheckIdenticalMotion :497, xPredInterBi :596,
// xWeightedAverage :700 -> TComYuv::addAvg :336; weighted prediction off)
HM_DEV inline void motion_compensation_pu(Shared *e, int cuZ, Rect r, Pel *dst, int list = 2)
{
  const InterPic *s = e->fb.ip; const InterMeta *m = e->im;
  if (list != 2) { pred_inter_uni(e, cuZ, r, list, dst, 0); return; }
  const int z = rect_z(r), r0 = m->refIdx[0][z], r1 = m->refIdx[1][z];
  if (r0 >= 0 && r1 >= 0) {
    if (s->ref[0][r0].poc == s->ref[1][r1].poc && m->mv[0][z].x == m->mv[1][z].x && m->mv[0][z].y == m->mv[1][z].y) { pred_inter_uni(e, cuZ, r, 0, dst, 0); return; }
    Pel *p0 = e->ws->yuvPred[0], *p1 = e->ws->yuvPred[1];
    pred_inter_uni(e, cuZ, r, 0, p0, 1); pred_inter_uni(e, cuZ, r, 1, p1, 1);
    const int bd = e->bitDepth, shiftNum = ((14 - bd) > 2 ? (14 - bd) : 2) + 1, offset = (1 << (shiftNum - 1)) + 2 * HM_IF_OFFS, maxv = (1 << bd) - 1;
    for (int c = 0; c < 3; c++) {
      const int st = HM_PLANE_STRIDE(c), sh = c ? 1 : 0, o0 = HM_PLANE_OFF(c) + (r.y >> sh) * st + (r.x >> sh), cw = r.w >> sh, ch = r.h >> sh;
      HM_PAR_FOR_XY(x, y, cw, cw * ch) { const int o = o0 + y * st + x; dst[o] = (Pel)hm_clip3(0, maxv, (p0[o] + p1[o] + offset) >> shiftNum); }
    }
    HM_SYNC();
    return;
  }
  pred_inter_uni(e, cuZ, r, r0 >= 0 ? 0 : 1, dst, 0);
}

// AMVP predictor choice (xEstimateMvPredAMVP :3571, xGetTemplateCost :3771, xCheckBestMVP :3725)
HM_DEV inline uint32_t template_cost(Shared *e, int cuZ, Rect r, MvD cand, int list, int refIdx)
{
  const InterPic *s = e->fb.ip;
  const int rz = hm_z2r(cuZ), cuX = e->ctuX * 64 + (rz & 15) * 4, cuY = e->ctuY * 64 + (rz >> 4) * 4;
  const int px = e->ctuX * 64 + r.x, py = e->ctuY * 64 + r.y;
  const int inLds = r.w * r.h <= (int)(sizeof(e->u.bufB) / sizeof(Pel));
  Pel *blk = inLds ? (Pel *)e->u.bufB : e->ws->mcBlk; const int bs = inLds ? r.w : 64;
  pred_inter_blk(e, 0, &s->ref[list][refIdx], px, py, clip_mv(e, cand, cuX, cuY), r.w, r.h, blk, bs);
  const uint32_t sad = dist_sad_rect(e->fb.org[0] + (ptrdiff_t)py * e->stride[0] + px, e->stride[0], blk, bs, r.w, r.h, 0, e->bitDepth);
  HM_TRACE(e, 9, ((uint32_t)(uint16_t)cand.x << 16) | (uint16_t)cand.y, sad, 0.0);
  const double t = floor(((double)1 * (double)s->lambdaMotionSAD) + 0.5) / 65536.0;
  return (uint32_t)floor((double)sad + t);
}
HM_DEV inline MvD estimate_mvp_amvp(Shared *e, int cuZ, int cuDepth, int partSize, int puIdx, int list, int refIdx, AmvpInfo *info, int *bestIdx, uint32_t *distBiP)
{
  const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
  fill_mvp_cand(e, cuZ, cuDepth, partSize, puIdx, list, refIdx, info);
  *bestIdx = 0;
  MvD best = info->cand[0];
  if (info->n > 1) {                                            // always: the list is padded to AMVP_MAX_NUM_CANDS
    uint32_t bestCost = 0xffffffffu;
    for (int i = 0; i < info->n; i++) {
      const uint32_t c = template_cost(e, cuZ, r, info->cand[i], list, refIdx);
      if (bestCost > c) { bestCost = c; best = info->cand[i]; *bestIdx = i; *distBiP = c; }   // (*puiDistBiP) = uiTmpCost :3626
    }
  }
  pu_set_mvp(e, r, list, *bestIdx, info->n);
  return best;
}
HM_DEV inline void check_best_mvp(Shared *e, const AmvpInfo *info, MvD mv, MvD *mvPred, int *mvpIdx, uint32_t *bits, uint32_t *cost)
{
  if (info->n < 2) return;
  e->mcost = e->fb.ip->lambdaMotionSAD; e->costScale = 0;
  int bestIdx = *mvpIdx;
  e->mvPredictor = *mvPred;
  const int orgBits = (int)mc_bits(e, mv.x, mv.y) + 1;
  int bestBits = orgBits;
  for (int i = 0; i < info->n; i++) {
    if (i == *mvpIdx) continue;
    e->mvPredictor = info->cand[i];
    const int b = (int)mc_bits(e, mv.x, mv.y) + 1;
    if (b < bestBits) { bestBits = b; bestIdx = i; }
  }
  if (bestIdx != *mvpIdx) {
    *mvPred = info->cand[bestIdx]; *mvpIdx = bestIdx;
    const uint32_t org = *bits;
    *bits = org - orgBits + bestBits;
    *cost = (*cost - mc_cost32(e, org)) + mc_cost32(e, *bits);
  }
}

// xMergeEstimation :2987 (+ xGetInterPredictionError :2952); result in e->mrg*
HM_DEV HM_NOINLINE void merge_estimation(Shared *e, int cuZ, int cuDepth, int partSize, int puIdx)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); puIdx = HM_UNI(puIdx);
  const InterPic *s = e->fb.ip;
  const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
  MergeList *ml = &e->ml;
  merge_candidates(e, cuZ, cuDepth, partSize, puIdx, ml);
  if ((64 >> cuDepth) == 8 && (r.w < 8 || r.h < 8))             // xRestrictBipredMergeCand :3047
    for (int i = 0; i < ml->num; i++) if (ml->dir[i] == 3) { ml->dir[i] = 1; ml->f[i][1].mv.x = ml->f[i][1].mv.y = 0; ml->f[i][1].ref = -1; }
  uint32_t cost = 0xffffffffu;
  const int px = e->ctuX * 64 + r.x, py = e->ctuY * 64 + r.y;
  for (int c = 0; c < ml->num; c++) {
    pu_set_motion(e, r, 0, ml->f[c][0].mv, ml->f[c][0].ref);
    pu_set_motion(e, r, 1, ml->f[c][1].mv, ml->f[c][1].ref);
    motion_compensation_pu(e, cuZ, r, e->ws->tmpPred);
    uint32_t d = dist_hads_rect(e->fb.org[0] + (ptrdiff_t)py * e->stride[0] + px, e->stride[0], e->ws->tmpPred + r.y * 64 + r.x, 64, r.w, r.h, e->bitDepth);
    uint32_t b = (uint32_t)c + 1;
    if (c == s->maxMergeCand - 1) b--;
    d += mc_cost32(e, b);
    if (d < cost) { cost = d; e->mrgField[0] = ml->f[c][0]; e->mrgField[1] = ml->f[c][1]; e->mrgDir = ml->dir[c]; e->mrgIdx = c; }
  }
  e->mrgCost = cost;
  HM_TRACE(e, 3, e->mrgIdx, cost, (double)ml->num);
}

// TEncSearch::predInterSearch :3075-3567 (P and B slices; FEN: one bi-prediction iteration).  The per-(list, refIdx) arrays of
// the reference live in ws->ps; every lane holds the same scalars.
HM_DEV inline void get_blk_bits(int partSize, int isP, int puIdx, uint32_t lastMode, uint32_t blkBit[3])
{ // xGetBlkBits :3667-3713
  if (partSize == SIZE_2Nx2N || partSize == SIZE_NxN) { blkBit[0] = isP ? 1 : 3; blkBit[1] = 3; blkBit[2] = 5; return; }
  if (isP) { blkBit[0] = 3; blkBit[1] = 0; blkBit[2] = 0; return; }
  const int isHor = partSize == SIZE_2NxN || partSize == SIZE_2NxnU || partSize == SIZE_2NxnD;
  if (puIdx == 0) { blkBit[0] = 0; blkBit[1] = (lastMode == 0 && !isHor) ? 2 : 0; blkBit[2] = lastMode == 0 ? 3 : 0; return; }
  // aauiMbBits[1][lastMode]: 2NxN {5,7,7},{7,5,7},{6,6,6}; Nx2N {5,7,7},{5,5,7},{6,6,6}
  if (lastMode == 0) { blkBit[0] = 5; blkBit[1] = 7; blkBit[2] = 7; }
  else if (lastMode == 1) { blkBit[0] = isHor ? 7 : 5; blkBit[1] = 5; blkBit[2] = 7; }
  else { blkBit[0] = blkBit[1] = blkBit[2] = 6; }
}
HM_DEV HM_NOINLINE void pred_inter_search(Shared *e, int cuZ, int cuDepth, int partSize, int useMRG)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize); useMRG = HM_UNI(useMRG);
  const InterPic *s = e->fb.ip; InterMeta *m = e->im;
  WorkSpace::PuSearch *ps = &e->ws->ps;
  const int isP = s->sliceType == HM_P_SLICE, numPredDir = isP ? 1 : 2, mvdL1Zero = s->mvdL1Zero;
  const int numPart = num_parts_of(partSize), cuW = 64 >> cuDepth;
  MvD zero; zero.x = zero.y = 0;
  // declared outside the PU loop in the reference (:3098-3135): carried from PU 0 to PU 1
  MvD cMv[2], cMvBi[2]; cMv[0] = cMv[1] = cMvBi[0] = cMvBi[1] = zero;
  int refIdx[2] = {0, 0}, refIdxBi[2] = {0, 0};
  uint32_t lastMode = 0, biPDistTemp = 0xffffffffu;
  int bestBiPRefIdxL1 = 0, bestBiPMvpL1 = 0;
  for (int puIdx = 0; puIdx < numPart; puIdx++) {
    const Rect r = pu_rect(cuZ, cuDepth, partSize, puIdx);
    uint32_t cost[2] = {0xffffffffu, 0xffffffffu}, costBi = 0xffffffffu, costTemp, bits[3] = {0, 0, 0}, bitsTemp, bestBiPDist = 0xffffffffu;
    MvD mvValidList1 = zero; int refIdxValidList1 = 0; uint32_t bitsValidList1 = 0xffffffffu, costValidList1 = 0xffffffffu;
    uint32_t mbBits[3];
    get_blk_bits(partSize, isP, puIdx, lastMode, mbBits);
    const int testNormalMC = !(useMRG && cuW > 8 && numPart == 2);
    if (testNormalMC) {
      for (int list = 0; list < numPredDir; list++) {               // uni-directional prediction :3166-3259
        const int numRef = s->numRefIdx[list];
        for (int ri = 0; ri < numRef; ri++) {
          bitsTemp = mbBits[list];
          if (numRef > 1) { bitsTemp += ri + 1; if (ri == numRef - 1) bitsTemp--; }
          AmvpInfo *info = &e->amvp; int mvpIdx;
          HM_PROF_BEGIN(e, PR_AMVP);
          MvD mvPred = estimate_mvp_amvp(e, cuZ, cuDepth, partSize, puIdx, list, ri, info, &mvpIdx, &biPDistTemp);
          HM_PROF_END(e, PR_AMVP);
          ps->cand[list][ri][0] = info->cand[0]; ps->cand[list][ri][1] = info->cand[1]; ps->mvpNum[list][ri] = (int8_t)info->n;
          if (mvdL1Zero && list == 1 && biPDistTemp < bestBiPDist) { bestBiPDist = biPDistTemp; bestBiPMvpL1 = mvpIdx; bestBiPRefIdxL1 = ri; }
          bitsTemp += 1;                                            // m_auiMVPIdxCost[idx][AMVP_MAX_NUM_CANDS]
          MvD mvTemp;
          const int r0 = list == 1 ? s->list1ToList0[ri] : -1;
          if (r0 >= 0) {                                            // GPB_SIMPLE_UNI :3190-3204: same picture as a list-0 entry
            mvTemp = ps->mvTemp[0][r0];
            costTemp = ps->costTempL0[r0];
            costTemp -= mc_cost32(e, ps->bitsTempL0[r0]);
            e->mvPredictor = mvPred;
            bitsTemp += mc_bits(e, mvTemp.x, mvTemp.y);
            costTemp += mc_cost32(e, bitsTemp);
          } else {
            motion_estimation(e, cuZ, cuDepth, partSize, puIdx, mvPred.x, mvPred.y, (list << 4) | ri, bitsTemp, 0, 0, 0);
            mvTemp = e->outMv; bitsTemp = e->outBits; costTemp = e->outDist;
          }
          check_best_mvp(e, info, mvTemp, &mvPred, &mvpIdx, &bitsTemp, &costTemp);
          ps->mvTemp[list][ri] = mvTemp; ps->mvPred[list][ri] = mvPred; ps->mvpIdx[list][ri] = (int8_t)mvpIdx;
          if (list == 0) { ps->costTempL0[ri] = costTemp; ps->bitsTempL0[ri] = bitsTemp; }
          if (costTemp < cost[list]) { cost[list] = costTemp; bits[list] = bitsTemp; cMv[list] = mvTemp; refIdx[list] = ri; }
          if (list == 1 && costTemp < costValidList1 && r0 < 0) { costValidList1 = costTemp; bitsValidList1 = bitsTemp; mvValidList1 = mvTemp; refIdxValidList1 = ri; }
        }
      }
      // bi-directional prediction :3262-3416; isBipredRestriction :2902: none for 8x4 / 4x8
      if (!isP && !(cuW == 8 && (r.w < 8 || r.h < 8))) {
        cMvBi[0] = cMv[0]; cMvBi[1] = cMv[1]; refIdxBi[0] = refIdx[0]; refIdxBi[1] = refIdx[1];
        for (int l = 0; l < 2; l++) for (int i = 0; i < s->numRefIdx[l]; i++) { ps->mvPredBi[l][i] = ps->mvPred[l][i]; ps->mvpIdxBi[l][i] = ps->mvpIdx[l][i]; }
        uint32_t motBits[2];
        if (mvdL1Zero) {
          { InterMeta *mm = e->im; HM_PU_FOR(r, z) mm->mvpIdx[1][z] = (int8_t)bestBiPMvpL1; HM_SYNC(); }
          ps->mvpIdxBi[1][bestBiPRefIdxL1] = (int8_t)bestBiPMvpL1;
          ps->mvPredBi[1][bestBiPRefIdxL1] = ps->cand[1][bestBiPRefIdxL1][bestBiPMvpL1];
          cMvBi[1] = ps->mvPredBi[1][bestBiPRefIdxL1]; refIdxBi[1] = bestBiPRefIdxL1;
          pu_set_motion(e, r, 1, cMvBi[1], refIdxBi[1]);
          motion_compensation_pu(e, cuZ, r, e->ws->yuvPred[1], 1);
          motBits[0] = bits[0] - mbBits[0]; motBits[1] = mbBits[1];
          if (s->numRefIdx[1] > 1) { motBits[1] += bestBiPRefIdxL1 + 1; if (bestBiPRefIdxL1 == s->numRefIdx[1] - 1) motBits[1]--; }
          motBits[1] += 1;
          bits[2] = mbBits[2] + motBits[0] + motBits[1];
          ps->mvTemp[1][bestBiPRefIdxL1] = cMvBi[1];
        } else { motBits[0] = bits[0] - mbBits[0]; motBits[1] = bits[1] - mbBits[1]; bits[2] = mbBits[2] + motBits[0] + motBits[1]; }
        // FEN (getUseFastEnc) or mvd_l1_zero: one iteration, refining the list with the larger uni-directional cost
        int list = cost[0] <= cost[1] ? 1 : 0;
        if (!mvdL1Zero) { pu_set_motion(e, r, 1 - list, cMv[1 - list], refIdx[1 - list]); motion_compensation_pu(e, cuZ, r, e->ws->yuvPred[1 - list], 1 - list); }
        else list = 0;
        const int numRef = s->numRefIdx[list];
        for (int ri = 0; ri < numRef; ri++) {
          bitsTemp = mbBits[2] + motBits[1 - list];
          if (numRef > 1) { bitsTemp += ri + 1; if (ri == numRef - 1) bitsTemp--; }
          bitsTemp += 1;
          MvD mvPred = ps->mvPredBi[list][ri]; const MvD in = ps->mvTemp[list][ri]; int mvpIdx = ps->mvpIdxBi[list][ri];
          motion_estimation(e, cuZ, cuDepth, partSize, puIdx, mvPred.x, mvPred.y, (list << 4) | ri, bitsTemp, 1, in.x, in.y);
          const MvD mvTemp = e->outMv; bitsTemp = e->outBits; costTemp = e->outDist;
          AmvpInfo *info = &e->amvp; info->n = ps->mvpNum[list][ri]; info->cand[0] = ps->cand[list][ri][0]; info->cand[1] = ps->cand[list][ri][1];
          check_best_mvp(e, info, mvTemp, &mvPred, &mvpIdx, &bitsTemp, &costTemp);
          ps->mvTemp[list][ri] = mvTemp; ps->mvPredBi[list][ri] = mvPred; ps->mvpIdxBi[list][ri] = (int8_t)mvpIdx;
          if (costTemp < costBi) {
            cMvBi[list] = mvTemp; refIdxBi[list] = ri; costBi = costTemp;
            motBits[list] = bitsTemp - mbBits[2] - motBits[1 - list]; bits[2] = bitsTemp;
          }
        }
      }
    }
    // clear the motion field, then set the winner :3420-3493
    pu_set_motion(e, r, 0, zero, -1); pu_set_motion(e, r, 1, zero, -1);
    pu_set_mvd(e, r, 0, zero); pu_set_mvd(e, r, 1, zero);
    pu_set_mvp(e, r, 0, -1, -1); pu_set_mvp(e, r, 1, -1, -1);
    uint32_t meBits = 0;
    cMv[1] = mvValidList1; refIdx[1] = refIdxValidList1; bits[1] = bitsValidList1; cost[1] = costValidList1;
    if (testNormalMC) {
      if (costBi <= cost[0] && costBi <= cost[1]) {
        lastMode = 2;
        for (int l = 0; l < 2; l++) {
          pu_set_motion(e, r, l, cMvBi[l], refIdxBi[l]);
          const MvD p = ps->mvPredBi[l][refIdxBi[l]]; MvD mvd; mvd.x = (int16_t)(cMvBi[l].x - p.x); mvd.y = (int16_t)(cMvBi[l].y - p.y);
          pu_set_mvd(e, r, l, mvd);
          pu_set_mvp(e, r, l, ps->mvpIdxBi[l][refIdxBi[l]], ps->mvpNum[l][refIdxBi[l]]);
        }
        pu_set_u8(m->interDir, r, 3);
        meBits = bits[2];
      } else {
        const int l = cost[0] <= cost[1] ? 0 : 1;
        lastMode = (uint32_t)l;
        pu_set_motion(e, r, l, cMv[l], refIdx[l]);
        const MvD p = ps->mvPred[l][refIdx[l]]; MvD mvd; mvd.x = (int16_t)(cMv[l].x - p.x); mvd.y = (int16_t)(cMv[l].y - p.y);
        pu_set_mvd(e, r, l, mvd);
        pu_set_u8(m->interDir, r, 1 + l);
        pu_set_mvp(e, r, l, ps->mvpIdx[l][refIdx[l]], ps->mvpNum[l][refIdx[l]]);
        meBits = bits[l];
      }
    }
    if (partSize != SIZE_2Nx2N) {
      e->mcost = s->lambdaMotionSAD;
      uint32_t meCost = 0xffffffffu;
      if (testNormalMC) {
        motion_compensation_pu(e, cuZ, r, e->ws->tmpPred);
        const int px = e->ctuX * 64 + r.x, py = e->ctuY * 64 + r.y;
        const uint32_t err = dist_hads_rect(e->fb.org[0] + (ptrdiff_t)py * e->stride[0] + px, e->stride[0], e->ws->tmpPred + r.y * 64 + r.x, 64, r.w, r.h, e->bitDepth);
        meCost = err + mc_cost32(e, meBits);
      }
      const int z0 = rect_z(r);
      const int meDir = m->interDir[z0];
      const MvD meMv0 = m->mv[0][z0], meMv1 = m->mv[1][z0]; const int meRef0 = m->refIdx[0][z0], meRef1 = m->refIdx[1][z0];
      { HM_PROF_BEGIN(e, PR_MRG_EST); merge_estimation(e, cuZ, cuDepth, partSize, puIdx); HM_PROF_END(e, PR_MRG_EST); }
      if (e->mrgCost < meCost) {
        pu_set_u8(m->mrg, r, 1); pu_set_u8(m->mrgIdx, r, e->mrgIdx); pu_set_u8(m->interDir, r, e->mrgDir);
        pu_set_motion(e, r, 0, e->mrgField[0].mv, e->mrgField[0].ref); pu_set_motion(e, r, 1, e->mrgField[1].mv, e->mrgField[1].ref);
        pu_set_mvd(e, r, 0, zero); pu_set_mvd(e, r, 1, zero);
        pu_set_mvp(e, r, 0, -1, -1); pu_set_mvp(e, r, 1, -1, -1);
      } else {
        pu_set_u8(m->mrg, r, 0); pu_set_u8(m->interDir, r, meDir);
        pu_set_motion(e, r, 0, meMv0, meRef0); pu_set_motion(e, r, 1, meMv1, meRef1);
      }
    }
    { HM_PROF_BEGIN(e, PR_MC); motion_compensation_pu(e, cuZ, r, e->ws->pred); HM_PROF_END(e, PR_MC); }
  }
}

// ------------------------------------------------------------------------------------------------
// inter syntax on the estimator (TEncSbac.cpp:423-640,721-800; TEncEntropy.cpp:440-700)
// ------------------------------------------------------------------------------------------------
template <class C> HM_DEV inline void code_skip_flag(Shared *e, C *c, int z)
{ // codeSkipFlag :524, getCtxSkipFlag TComDataCU.cpp:1645
  int zz, ctx = 0;
  const int rz = hm_z2r(z), x4 = rz & 15, y4 = rz >> 4;
  int ca = nb_at(e, x4, y4, NB_LEFT, z, &zz); ctx = ca >= 0 ? imeta_of(e, ca)->skip[zz] : 0;
  ca = nb_at(e, x4, y4, NB_ABOVE, z, &zz); ctx += ca >= 0 ? imeta_of(e, ca)->skip[zz] : 0;
  enc_bin(e, c, C_SKIP + ctx, e->im->skip[z] ? 1 : 0);
}
template <class C> HM_DEV inline void code_merge_index(Shared *e, C *c, int z)
{ // codeMergeIndex :555
  const int idx = e->im->mrgIdx[z], num = e->fb.ip->maxMergeCand;
  if (num > 1)
    for (int ui = 0; ui < num - 1; ui++) {
      const int sym = ui == idx ? 0 : 1;
      if (ui == 0) enc_bin(e, c, C_MRG_IDX, sym); else enc_epv(c, (uint32_t)sym, 1);
      if (sym == 0) break;
    }
}
template <class C> HM_DEV inline void code_part_size_inter(Shared *e, C *c, int z, int depth)
{ // codePartSize :431-520, inter branch; AMP below the maximum depth
  const int size = e->meta.part[z], amp = depth < 3;
  switch (size) {
    case SIZE_2Nx2N: enc_bin(e, c, C_PART, 1); break;
    case SIZE_2NxN: case SIZE_2NxnU: case SIZE_2NxnD:
      enc_bin(e, c, C_PART, 0); enc_bin(e, c, C_PART + 1, 1);
      if (amp) { if (size == SIZE_2NxN) enc_bin(e, c, C_PART + 3, 1); else { enc_bin(e, c, C_PART + 3, 0); enc_epv(c, size == SIZE_2NxnU ? 0 : 1, 1); } }
      break;
    case SIZE_Nx2N: case SIZE_nLx2N: case SIZE_nRx2N:
      enc_bin(e, c, C_PART, 0); enc_bin(e, c, C_PART + 1, 0);
      if (amp) { if (size == SIZE_Nx2N) enc_bin(e, c, C_PART + 3, 1); else { enc_bin(e, c, C_PART + 3, 0); enc_epv(c, size == SIZE_nLx2N ? 0 : 1, 1); } }
      break;
    default: break;
  }
}
template <class C> HM_DEV inline void write_ep_ex_golomb(C *c, uint32_t symbol, uint32_t count)
{ // xWriteEpExGolomb :309-330: unary prefix, a zero, then the remainder in `count` bits
  uint32_t numBins = 0, bins = 0;
  while (symbol >= (1u << count)) { bins = 2 * bins + 1; numBins++; symbol -= 1u << count; count++; }
  bins = 2 * bins; numBins++;
  bins = (bins << count) | symbol; numBins += count;
  enc_epv(c, bins, (int)numBins);
}
template <class C> HM_DEV inline void code_inter_dir(Shared *e, C *c, int z, int cuDepth)
{ // codeInterDir :723-740; getCtxInterDir = depth (TComDataCU.cpp:1662)
  const int dir = e->im->interDir[z] - 1;
  if (e->meta.part[z] == SIZE_2Nx2N || (64 >> cuDepth) != 8) enc_bin(e, c, C_INTER_DIR + cuDepth, dir == 2 ? 1 : 0);
  if (dir < 2) enc_bin(e, c, C_INTER_DIR + 4, dir);
}
template <class C> HM_DEV inline void code_mvd(Shared *e, C *c, int z, int list)
{ // codeMvd :757
  if (e->fb.ip->mvdL1Zero && list == 1 && e->im->interDir[z] == 3) return;
  const int hor = e->im->mvd[list][z].x, ver = e->im->mvd[list][z].y;
  enc_bin(e, c, C_MVD, hor != 0); enc_bin(e, c, C_MVD, ver != 0);
  const uint32_t ha = (uint32_t)hm_abs(hor), va = (uint32_t)hm_abs(ver);
  if (hor != 0) enc_bin(e, c, C_MVD + 1, ha > 1);
  if (ver != 0) enc_bin(e, c, C_MVD + 1, va > 1);
  if (hor != 0) { if (ha > 1) write_ep_ex_golomb(c, ha - 2, 1); enc_epv(c, hor < 0, 1); }
  if (ver != 0) { if (va > 1) write_ep_ex_golomb(c, va - 2, 1); enc_epv(c, ver < 0, 1); }
}
template <class C> HM_DEV inline void code_ref_idx(Shared *e, C *c, int z, int list)
{ // codeRefFrmIdx :737
  int ref = e->im->refIdx[list][z];
  enc_bin(e, c, C_REF, ref == 0 ? 0 : 1);
  if (ref > 0) {
    const int num = e->fb.ip->numRefIdx[list] - 2;
    ref--;
    for (int ui = 0; ui < num; ui++) {
      const int sym = ui == ref ? 0 : 1;
      if (ui == 0) enc_bin(e, c, C_REF + 1, sym); else enc_epv(c, (uint32_t)sym, 1);
      if (sym == 0) break;
    }
  }
}
template <class C> HM_DEV inline void code_pu_wise(Shared *e, C *c, int cuZ, int cuDepth)
{ // TEncEntropy::encodePUWise :477
  const InterMeta *m = e->im; const int partSize = e->meta.part[cuZ], np = num_parts_of(partSize);
  for (int p = 0; p < np; p++) {
    const int z = rect_z(pu_rect(cuZ, cuDepth, partSize, p));
    enc_bin(e, c, C_MRG_FLAG, m->mrg[z] ? 1 : 0);
    if (m->mrg[z]) code_merge_index(e, c, z);
    else {
      if (e->fb.ip->sliceType == HM_B_SLICE) code_inter_dir(e, c, z, cuDepth);
      for (int l = 0; l < 2; l++) {
        if (e->fb.ip->numRefIdx[l] > 0) {
          if (e->fb.ip->numRefIdx[l] > 1 && (m->interDir[z] & (1 << l))) code_ref_idx(e, c, z, l);
          if (m->interDir[z] & (1 << l)) { code_mvd(e, c, z, l); enc_bin(e, c, C_MVP_IDX, m->mvpIdx[l][z]); }
        }
      }
    }
  }
}
HM_DEV inline int qt_root_cbf(const CtuMeta *m, int z) { return (m->cbf[0][z] & 1) || (m->cbf[1][z] & 1) || (m->cbf[2][z] & 1); }

// TEncEntropy::xEncodeTransform :222-412 for an inter CU (coefficients from the CTU arrays)
template <class C> HM_DEV inline void encode_transform_inter(Shared *e, C *c, const TU *root, int codeDqp)
{
  const CtuMeta *m = &e->meta;
  TuWalk &w = e->walkOuter; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    const int subdiv = m->tr[z] > t->trDepth;
    if (w.next[w.sp] < 0) {
      if (t->log2 > 5) { }
      else if (t->log2 == 2) { }
      else if (t->log2 == tr_min_size_in_cu(6 - t->cuDepth, 0)) { }
      else enc_bin(e, c, C_SUBDIV + (5 - t->log2), subdiv);
      const int first = t->trDepth == 0;
      for (int comp = 1; comp < 3; comp++)
        if (first || t->cCodeAll)
          if (first || ((m->cbf[comp][z] >> (t->trDepth - 1)) & 1)) code_qt_cbf(e, c, t, comp, subdiv == 0);
      if (!subdiv) {
        if (!(first && !(m->cbf[1][z] & 1) && !(m->cbf[2][z] & 1))) code_qt_cbf(e, c, t, 0, 1);
        if (codeDqp && (((m->cbf[0][z] | m->cbf[1][z] | m->cbf[2][z]) >> t->trDepth) & 1)) code_dqp_if_due(e, c);
        for (int comp = 0; comp < 3; comp++) {
          if (comp && !t->cW) continue;
          if (!((m->cbf[comp][z] >> t->trDepth) & 1)) continue;
          const int n = comp ? t->cW : (1 << t->log2);
          const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
          const TCoeff *coef = e->cc + HM_PLANE_OFF(comp) + (comp ? t->cOff : z * 16);
          code_coeff_nxn(e, c, coef, n, comp, SCAN_DIAG, m->ts[comp][zc]);
        }
        w.sp--; continue;
      }
      w.next[w.sp] = 0;
    }
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 1); w.next[w.sp + 1] = -1; w.sp++;
  }
}
// the whole inter CU: xAddSymbolBitsInter :5517 (codeDqp 0: "Bool codeDeltaQp = false") and xEncodeCU :1246-1290 (codeDqp 1: TEncCu::m_bEncodeDQP decides)
template <class C> HM_DEV HM_NOINLINE void encode_cu_syntax_inter(Shared *e, C *c, int cuZ, int cuDepth, int codeDqp)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); codeDqp = HM_UNI(codeDqp); c = hm_uni_ptr(c); HM_ASSUME_LDS(c);
  const CtuMeta *m = &e->meta; const InterMeta *im = e->im;
  code_skip_flag(e, c, cuZ);
  if (im->skip[cuZ]) { code_merge_index(e, c, cuZ); return; }
  enc_bin(e, c, C_PRED_MODE, 0);
  code_part_size_inter(e, c, cuZ, cuDepth);
  code_pu_wise(e, c, cuZ, cuDepth);
  if (!(im->mrg[cuZ] && m->part[cuZ] == SIZE_2Nx2N)) enc_bin(e, c, C_ROOT_CBF, qt_root_cbf(m, cuZ));
  if (!qt_root_cbf(m, cuZ)) return;
  const TU root = tu_root(e, cuZ, cuDepth);
  encode_transform_inter(e, c, &root, codeDqp);
}

// ------------------------------------------------------------------------------------------------
// inter residual quadtree (xEstimateResidualQT :4680-5306, xEncodeResidualQT :5308, xSetResidualQTData :5391)
// ------------------------------------------------------------------------------------------------
HM_DEV inline void encode_residual_qt(Shared *e, const TU *root, int comp /* 3 = flags pass */)
{
  const CtuMeta *m = &e->meta;
  TuWalk &w = e->walkInner; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ, trMode = m->tr[z], subdiv = t->trDepth != trMode;
    if (w.next[w.sp] < 0) {
      if (comp == 3) {
        if (t->log2 <= 5 && t->log2 > tr_min_size_in_cu(6 - t->cuDepth, 0)) enc_bin(e, &e->cur, C_SUBDIV + (5 - t->log2), subdiv);
        const int first = t->trDepth == 0;
        for (int ch = 1; ch < 3; ch++)
          if (first || t->cCodeAll)
            if (first || ((m->cbf[ch][z] >> (t->trDepth - 1)) & 1)) code_qt_cbf(e, &e->cur, t, ch, !subdiv);
        if (!subdiv) code_qt_cbf(e, &e->cur, t, 0, 1);
      }
      if (!subdiv) {
        if (comp != 3 && !(comp && !t->cW)) {
          const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
          if ((m->cbf[comp][z] >> trMode) & 1) {
            const int n = comp ? t->cW : (1 << t->log2);
            const TCoeff *coef = e->ws->qtCoef[5 - t->log2] + HM_PLANE_OFF(comp) + (comp ? t->cOff : z * 16);
            code_coeff_nxn(e, &e->cur, coef, n, comp, SCAN_DIAG, m->ts[comp][zc]);
          }
        }
        w.sp--; continue;
      }
      if (!(comp == 3 || ((m->cbf[comp][z] >> t->trDepth) & 1))) { w.sp--; continue; }
      w.next[w.sp] = 0;
    }
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
  }
}
HM_DEV inline void set_residual_qt_data(Shared *e, const TU *root, int spatial)
{
  const CtuMeta *m = &e->meta;
  TuWalk &w = e->walkInner; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    if (w.next[w.sp] < 0) {
      if (t->trDepth == m->tr[z]) {
        const int layer = 5 - t->log2;
        for (int comp = 0; comp < 3; comp++) {
          if (comp && !t->cW) continue;
          const int n = comp ? t->cW : (1 << t->log2), l2 = hm_log2(n), st = HM_PLANE_STRIDE(comp), po = HM_PLANE_OFF(comp), bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
          const int coded = (m->cbf[comp][t->cuZ + (comp ? t->cRelZ : t->relZ)] >> t->trDepth) & 1;      // an empty block: zeros (irq_check_full leaves its buffers as they were)
          if (spatial) { HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); e->ws->resiBest[po + (by + y) * st + bx + x] = coded ? e->ws->qtRec[layer][po + (by + y) * st + bx + x] : (Pel)0; } }
          else { const int off = po + (comp ? t->cOff : z * 16); HM_PAR_FOR(i, n * n) e->cc[off + i] = coded ? e->ws->qtCoef[layer][off + i] : 0; }
        }
        HM_SYNC();
        w.sp--; continue;
      }
      w.next[w.sp] = 0;
    }
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
  }
}

// full (unsplit) evaluation of one TU: every component, transform-skip trial for 4x4 blocks (:4725-5106)
HM_DEV HM_NOINLINE void irq_check_full(Shared *e, int sp)
{
  HM_ENTRY(e); sp = HM_UNI(sp);
  IrqFrame *f = &e->irq[sp]; const TU *t = &f->t;
  CtuMeta *m = &e->meta; WorkSpace *ws = e->ws;
  const int z = t->cuZ + t->relZ, fullDepth = t->cuDepth + t->trDepth, trMode = t->trDepth, layer = 5 - t->log2;
  const int minLog2 = tr_min_size_in_cu(6 - t->cuDepth, 0), bd = e->bitDepth;
  uint32_t singleDist = 0;
  const CabacHold root = cabac_hold(&e->cur);                       // == slot[fullDepth][CI_QT_TRAFO_ROOT]: the caller stored it from e->cur just before
  par_set8(m->tr + z, trMode, t->parts);
  for (int comp = 0; comp < 3; comp++) {
    f->absSum[comp] = 0; f->bestTS[comp] = 0;
    if (comp && !t->cW) continue;
    const int n = comp ? t->cW : (1 << t->log2), l2 = hm_log2(n), st = HM_PLANE_STRIDE(comp), po = HM_PLANE_OFF(comp);
    const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ), parts = comp ? t->cParts : t->parts;
    const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
    const Pel *resi = ws->resi + po + by * st + bx;
    Pel *rq = ws->qtRec[layer] + po + by * st + bx;
    TCoeff *coef = ws->qtCoef[layer] + po + (comp ? t->cOff : z * 16);
    const int cbfCtx = comp ? 5 + t->trDepth : (t->trDepth == 0 ? 10 : 0);      // 10: the root-cbf slot of rdoq's table (:2301)
    const int nModes = (n == 4) ? 2 : 1, tshift = 15 - bd - l2;
    double minCost = HM_MAX_DOUBLE; uint32_t compDist = 0;
    for (int tsMode = 0; tsMode < nModes; tsMode++) {
      const int isFirst = tsMode == 0, isOne = nModes == 1;
      par_set8(m->ts[comp] + zc, tsMode, parts);
      cabac_put(&e->cur, root);
      reset_bits(&e->cur);
      uint32_t currBits = 0, currDist = 0, nonCoeffBits = 0, nonCoeffDist = 0; double currCost = 0, nonCoeffCost = 0;
      if (!isOne && !isFirst) { HM_PAR_FOR(i, 16) { e->ws->tsCoef[comp][i] = coef[i]; e->ws->tsRec[comp][i] = rq[(i >> 2) * st + (i & 3)]; } HM_SYNC(); }
      uint32_t sqResi = 0;                                            // SSE of the residual against zero, for the "no coefficients" alternative below
      { const int shiftSse = (bd - 8) << 1;
        HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); const int r = resi[y * st + x]; e->bufA[y * HM_TSTRIDE + x] = tsMode ? (r << tshift) : r; sqResi += (uint32_t)((r * r) >> shiftSse); } }
      HM_SYNC();
      { HM_PROF_BEGIN(e, PR_IQ_FWD); if (!tsMode) fwd_transform(e, n, 0, bd); HM_PROF_END(e, PR_IQ_FWD); }
      int absSum;
      { HM_PROF_BEGIN(e, PR_IQ_RDOQ); absSum = rdoq_is_empty(e, n, comp) ? 0 : (int)HM_UCALL(rdoq(e, coef, n, comp, SCAN_DIAG, cbfCtx)); HM_PROF_END(e, PR_IQ_RDOQ); }
      par_set8(m->cbf[comp] + zc, (absSum > 0 ? 1 : 0) << trMode, parts);
      if (isFirst || absSum == 0) {
        uint32_t d = hm_wave_sum(sqResi);
        if (comp) d = (uint32_t)(e->fb.chromaWeight * (double)d);
        nonCoeffDist = d;
        enc_bin(e, &e->cur, C_QT_CBF + (comp ? 5 : 0) + (comp ? t->trDepth : (t->trDepth == 0 ? 1 : 0)), 0);
        nonCoeffBits = num_bits(&e->cur);
        nonCoeffCost = calc_rd_cost(e, nonCoeffBits, nonCoeffDist);
      }
      if (f->zero && isFirst) e->irqZeroDist += nonCoeffDist;
      if (absSum > 0) {
        if (isFirst) { cabac_put(&e->cur, root); reset_bits(&e->cur); }
        { HM_PROF_BEGIN(e, PR_IQ_BITS); code_qt_cbf(e, &e->cur, t, comp, 1);
        code_coeff_nxn(e, &e->cur, coef, n, comp, SCAN_DIAG, tsMode); HM_PROF_END(e, PR_IQ_BITS); }
        currBits = num_bits(&e->cur);
        { // xDeQuant + inverse transform (:1423-1545)
          const int rightShift = 6 - (tshift + e->fb.qpPer[comp != 0]);
          const int scale = HM_INV_QUANT_SCALES[e->fb.qpRem[comp != 0]];
          int tgt = 25 + rightShift; if (tgt > 16) tgt = 16;
          const int imin = -(1 << (tgt - 1)), imax = (1 << (tgt - 1)) - 1;
          HM_PAR_FOR(i, n * n) {
            const int y = i >> l2, x = i & (n - 1);
            const int cq = hm_clip3(imin, imax, coef[i]);
            int v;
            if (rightShift > 0) v = (cq * scale + (1 << (rightShift - 1))) >> rightShift;
            else v = (int)((unsigned)(cq * scale) << (-rightShift));
            e->bufA[y * HM_TSTRIDE + x] = hm_clip3(-32768, 32767, v);
          }
          HM_SYNC();
          if (!tsMode) inv_transform(e, n, 0, bd);
          const int off = tshift == 0 ? 0 : (1 << (tshift - 1));
          HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); const int v = e->bufA[y * HM_TSTRIDE + x]; rq[y * st + x] = (Pel)(tsMode ? ((v + off) >> tshift) : v); }
          HM_SYNC();
        }
        uint32_t d = dist_sse(rq, st, resi, st, n, bd);
        if (comp) d = (uint32_t)(e->fb.chromaWeight * (double)d);
        currDist = d;
        currCost = calc_rd_cost(e, currBits, currDist);
      } else if (tsMode == 1) currCost = HM_MAX_DOUBLE;
      else { currBits = nonCoeffBits; currDist = nonCoeffDist; currCost = nonCoeffCost; }
      if (currCost < minCost || (tsMode == 1 && currCost == minCost)) {
        if (isFirst && (nonCoeffCost < currCost || absSum == 0)) { absSum = 0; currBits = nonCoeffBits; currDist = nonCoeffDist; currCost = nonCoeffCost; }
        // (an empty block's levels and reconstructed residual are not cleared here: nothing reads them while its cbf is 0, and
        // set_residual_qt_data writes zeros for such a block instead of copying)
        f->absSum[comp] = (uint32_t)absSum; compDist = currDist; minCost = currCost; f->bestTS[comp] = (uint8_t)tsMode;
        HM_SYNC();
      } else {
        HM_PAR_FOR(i, 16) { coef[i] = e->ws->tsCoef[comp][i]; rq[(i >> 2) * st + (i & 3)] = e->ws->tsRec[comp][i]; }
        HM_SYNC();
      }
      (void)currBits;
    }
    par_set8(m->ts[comp] + zc, f->bestTS[comp], parts);
    par_set8(m->cbf[comp] + zc, (f->absSum[comp] > 0 ? 1 : 0) << trMode, parts);
    singleDist += compDist;
  }
  cabac_put(&e->cur, root);
  reset_bits(&e->cur);
  if (t->log2 > minLog2) enc_bin(e, &e->cur, C_SUBDIV + (5 - t->log2), 0);
  for (int ch = 0; ch < 3; ch++) { const int comp = (ch + 1 == 3) ? 0 : ch + 1; if (!(comp && !t->cW)) code_qt_cbf(e, &e->cur, t, comp, 1); }
  for (int comp = 0; comp < 3; comp++) {
    if (comp && !t->cW) continue;
    const int n = comp ? t->cW : (1 << t->log2), zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
    if ((m->cbf[comp][zc] >> trMode) & 1) code_coeff_nxn(e, &e->cur, ws->qtCoef[layer] + HM_PLANE_OFF(comp) + (comp ? t->cOff : z * 16), n, comp, SCAN_DIAG, m->ts[comp][zc]);
  }
  f->singleDist = singleDist;
  f->singleBits = num_bits(&e->cur);
  f->singleCost = calc_rd_cost(e, f->singleBits, f->singleDist);
  HM_TRACE(e, 8, f->singleBits, f->singleDist, f->singleCost);
}

// results in e->outRdCost (sum of costs), e->outBits, e->outDist, e->irqZeroDist
HM_DEV HM_NOINLINE void estimate_residual_qt(Shared *e, TU rootv)
{
  HM_ENTRY(e); rootv = hm_uni_struct(rootv);
  CtuMeta *m = &e->meta; WorkSpace *ws = e->ws;
  IrqFrame *fr = e->irq; int sp = 0;
  fr[0].t = rootv; fr[0].phase = 0; fr[0].zero = 1;
  e->irqZeroDist = 0;
  double retCost = 0; uint32_t retBits = 0, retDist = 0;
  while (sp >= 0) {
    IrqFrame *f = &fr[sp]; const TU *t = &f->t;
    const int z = t->cuZ + t->relZ, fullDepth = t->cuDepth + t->trDepth, trMode = t->trDepth;
    if (f->phase == 0) {
      const int minLog2 = tr_min_size_in_cu(6 - t->cuDepth, 0);
      f->checkFull = t->log2 <= 5; f->checkSplit = t->log2 > minLog2;
      f->singleCost = HM_MAX_DOUBLE; f->singleBits = 0; f->singleDist = 0;
      cabac_copy(&ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
      if (f->checkFull) { HM_PROF_BEGIN(e, PR_IQ_FULL); irq_check_full(e, sp); HM_PROF_END(e, PR_IQ_FULL); }
      if (!f->checkSplit) { retCost = f->singleCost; retBits = f->singleBits; retDist = f->singleDist; sp--; continue; }
      if (f->checkFull) { cabac_copy(&ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_TEST)], &e->cur); cabac_copy(&e->cur, &ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]); }
      f->subCost = 0.0; f->subBits = 0; f->subDist = 0; f->child = 0;
      for (int comp = 0; comp < 3; comp++) f->bestCBF[comp] = (comp && !t->cW) ? 0 : ((m->cbf[comp][z] >> trMode) & 1);
      f->phase = 1;
    }
    if (f->phase == 1) {
      if (f->child < 4) {
        fr[sp + 1].t = tu_child(t, f->child, 0); fr[sp + 1].phase = 0; fr[sp + 1].zero = f->checkFull ? 0 : f->zero;
        f->child++; f->phase = 2; sp++; continue;
      }
      const int q = t->parts >> 2;
      uint32_t cbfAny = 0;
      for (int comp = 0; comp < 3; comp++) {
        uint32_t yuv = 0;
        for (int ui = 0; ui < 4; ui++) yuv |= (m->cbf[comp][z + ui * q] >> (trMode + 1)) & 1;
        HM_PAR_FOR(ui, 4 * q) m->cbf[comp][z + ui] |= (uint8_t)(yuv << trMode);
        cbfAny |= yuv;
      }
      HM_SYNC();
      cabac_copy(&e->cur, &ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]);
      reset_bits(&e->cur);
      { HM_PROF_BEGIN(e, PR_IQ_ENC); encode_residual_qt(e, t, 3);
      for (int comp = 0; comp < 3; comp++) encode_residual_qt(e, t, comp); HM_PROF_END(e, PR_IQ_ENC); }
      const uint32_t subdivBits = num_bits(&e->cur);
      const double subdivCost = calc_rd_cost(e, subdivBits, f->subDist);
      if (!f->checkFull || (cbfAny && subdivCost < f->singleCost)) { retCost = subdivCost; retBits = subdivBits; retDist = f->subDist; }
      else {
        retCost = f->singleCost; retBits = f->singleBits; retDist = f->singleDist;
        par_set8(m->tr + z, trMode, t->parts);
        for (int comp = 0; comp < 3; comp++) {
          if (comp && !t->cW) continue;
          const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ), parts = comp ? t->cParts : t->parts;
          par_set8(m->cbf[comp] + zc, (int)(f->bestCBF[comp] << trMode), parts);
          par_set8(m->ts[comp] + zc, f->bestTS[comp], parts);
        }
        cabac_copy(&e->cur, &ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_TEST)]);
      }
      sp--; continue;
    }
    if (f->phase == 2) { f->subCost += retCost; f->subBits += retBits; f->subDist += retDist; f->phase = 1; continue; }
  }
  e->outRdCost = retCost; e->outBits = retBits; e->outDist = retDist;
}

// TEncSearch::encodeResAndCalcRdInterCU :4435-4676; results in e->outCost / outBits / outDist
HM_DEV HM_NOINLINE void encode_res_and_calc_rd_inter(Shared *e, int cuZ, int cuDepth, int skipRes)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); skipRes = HM_UNI(skipRes);
  CtuMeta *m = &e->meta; InterMeta *im = e->im; WorkSpace *ws = e->ws;
  const int n = 64 >> cuDepth, l2 = 6 - cuDepth, parts = 256 >> (2 * cuDepth), bd = e->bitDepth;
  const int rz = hm_z2r(cuZ), x0 = (rz & 15) * 4, y0 = (rz >> 4) * 4;
  uint32_t bits = 0, dist = 0;
  if (skipRes) {
    par_set8(im->skip + cuZ, 1, parts);
    for (int c = 0; c < 3; c++) {
      const int sh = c ? 1 : 0, st = HM_PLANE_STRIDE(c), po = HM_PLANE_OFF(c), nn = n >> sh, ll = l2 - sh;
      const Pel *org = e->fb.org[c] + (e->ctuY * st + (y0 >> sh)) * e->stride[c] + e->ctuX * st + (x0 >> sh);
      const int o0 = po + (y0 >> sh) * st + (x0 >> sh);
      HM_PAR_FOR(i, nn * nn) { const int y = i >> ll, x = i & (nn - 1); ws->reco[o0 + y * st + x] = ws->pred[o0 + y * st + x]; }
      HM_SYNC();
      uint32_t d = dist_sse(ws->reco + o0, st, org, e->stride[c], nn, bd);
      if (c) d = (uint32_t)(e->fb.chromaWeight * (double)d);
      dist += d;
    }
    cabac_copy(&e->cur, &ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
    reset_bits(&e->cur);
    code_skip_flag(e, &e->cur, cuZ);
    code_merge_index(e, &e->cur, cuZ);
    bits = num_bits(&e->cur);
    e->outBits = bits; e->outDist = dist; e->outCost = calc_rd_cost(e, bits, dist);
    cabac_copy(&ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)], &e->cur);
    for (int c = 0; c < 3; c++) par_set8(m->cbf[c] + cuZ, 0, parts);
    par_set8(m->tr + cuZ, 0, parts);
    return;
  }
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, st = HM_PLANE_STRIDE(c), po = HM_PLANE_OFF(c), nn = n >> sh, ll = l2 - sh;
    const Pel *org = e->fb.org[c] + (e->ctuY * st + (y0 >> sh)) * e->stride[c] + e->ctuX * st + (x0 >> sh);
    const int o0 = po + (y0 >> sh) * st + (x0 >> sh);
    HM_PAR_FOR(i, nn * nn) { const int y = i >> ll, x = i & (nn - 1); ws->resi[o0 + y * st + x] = (Pel)(org[y * e->stride[c] + x] - ws->pred[o0 + y * st + x]); }
  }
  HM_SYNC();
  const TU root = tu_root(e, cuZ, cuDepth);
  cabac_copy(&e->cur, &ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
  { HM_PROF_BEGIN(e, PR_IRQ); estimate_residual_qt(e, root); HM_PROF_END(e, PR_IRQ); }
  const double dCost = e->outRdCost; dist = e->outDist;
  HM_TRACE(e, 6, e->outBits, e->outDist, dCost);
  const uint32_t zeroDist = e->irqZeroDist;
  reset_bits(&e->cur);
  enc_bin(e, &e->cur, C_ROOT_CBF, 0);
  const uint32_t zeroResiBits = num_bits(&e->cur);
  const double zeroCost = calc_rd_cost(e, zeroResiBits, zeroDist);
  if (zeroCost < dCost) {
    dist = zeroDist;
    par_set8(m->tr + cuZ, 0, parts);
    for (int c = 0; c < 3; c++) { par_set8(m->cbf[c] + cuZ, 0, parts); par_set8(m->ts[c] + cuZ, 0, parts); }
    HM_PAR_FOR(i, parts * 16) e->cc[cuZ * 16 + i] = 0;
    HM_PAR_FOR(i, parts * 4) { e->cc[4096 + cuZ * 4 + i] = 0; e->cc[5120 + cuZ * 4 + i] = 0; }
    HM_SYNC();
  } else set_residual_qt_data(e, &root, 0);
  cabac_copy(&e->cur, &ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
  if (im->mrg[cuZ] && m->part[cuZ] == SIZE_2Nx2N && !qt_root_cbf(m, cuZ)) par_set8(im->skip + cuZ, 1, parts);
  reset_bits(&e->cur);
  encode_cu_syntax_inter(e, &e->cur, cuZ, cuDepth, 0);
  bits = num_bits(&e->cur);
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, st = HM_PLANE_STRIDE(c), po = HM_PLANE_OFF(c), nn = n >> sh, ll = l2 - sh, o0 = po + (y0 >> sh) * st + (x0 >> sh);
    HM_PAR_FOR(i, nn * nn) { const int y = i >> ll, x = i & (nn - 1); ws->resiBest[o0 + y * st + x] = 0; }
  }
  HM_SYNC();
  if (qt_root_cbf(m, cuZ)) set_residual_qt_data(e, &root, 1);
  cabac_copy(&ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)], &e->cur);
  const int maxv = (1 << bd) - 1;
  uint32_t distBest = 0;
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, st = HM_PLANE_STRIDE(c), po = HM_PLANE_OFF(c), nn = n >> sh, ll = l2 - sh, o0 = po + (y0 >> sh) * st + (x0 >> sh);
    const Pel *org = e->fb.org[c] + (e->ctuY * st + (y0 >> sh)) * e->stride[c] + e->ctuX * st + (x0 >> sh);
    HM_PAR_FOR(i, nn * nn) { const int y = i >> ll, x = i & (nn - 1), o = o0 + y * st + x; ws->reco[o] = (Pel)hm_clip3(0, maxv, ws->pred[o] + ws->resiBest[o]); }
    HM_SYNC();
    uint32_t d = dist_sse(ws->reco + o0, st, org, e->stride[c], nn, bd);
    if (c) d = (uint32_t)(e->fb.chromaWeight * (double)d);
    distBest += d;
  }
  e->outCost = calc_rd_cost(e, bits, distBest); e->outBits = bits; e->outDist = distBest;
  if (im->skip[cuZ]) for (int c = 0; c < 3; c++) par_set8(m->cbf[c] + cuZ, 0, parts);
}

// ------------------------------------------------------------------------------------------------
// CU level (TEncCu::xCheckRDCostMerge2Nx2N :1406, xCheckRDCostInter :1532, deriveTestModeAMP :386)
// The caller's frame keeps the best cost; these leave the candidate result in e->outCost / outBits / outDist.
// ------------------------------------------------------------------------------------------------
HM_DEV inline void init_est_data_inter(Shared *e, int cuZ, int cuDepth)
{ // the inter part of TComDataCU::initEstData
  InterMeta *im = e->im; const int parts = 256 >> (2 * cuDepth);
  HM_PAR_FOR(i, parts) {
    const int z = cuZ + i;
    im->skip[z] = 0; im->mrg[z] = 0; im->mrgIdx[z] = 0; im->interDir[z] = 0;
    for (int l = 0; l < 2; l++) { im->mv[l][z].x = im->mv[l][z].y = 0; im->mvd[l][z].x = im->mvd[l][z].y = 0; im->refIdx[l][z] = -1; im->mvpIdx[l][z] = -1; im->mvpNum[l][z] = -1; }
  }
  HM_SYNC();
}
HM_DEV inline void imeta_copy_range(InterMeta *d, const InterMeta *s, int z0, int parts)
{
  HM_PAR_FOR(i, parts) {
    const int z = z0 + i;
    d->skip[z] = s->skip[z]; d->mrg[z] = s->mrg[z]; d->mrgIdx[z] = s->mrgIdx[z]; d->interDir[z] = s->interDir[z];
    for (int l = 0; l < 2; l++) { d->mv[l][z] = s->mv[l][z]; d->mvd[l][z] = s->mvd[l][z]; d->refIdx[l][z] = s->refIdx[l][z]; d->mvpIdx[l][z] = s->mvpIdx[l][z]; d->mvpNum[l][z] = s->mvpNum[l][z]; }
  }
  HM_SYNC();
}
