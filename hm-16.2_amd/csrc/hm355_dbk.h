// hm355 -- deblocking filter (TComLoopFilter, TLibCommon/TComLoopFilter.cpp): the step after compressSlice that turns the
// reconstruction into a reference picture.  Data-parallel and HBM-bound: one lane per 4-sample edge
// segment, lanes laid along the picture row so that a wavefront's accesses to a row are contiguous (16 B per lane for vertical
// luma edges, 8 B per lane for horizontal ones).  Filtering is in place: within one direction the samples an edge reads
// (4 each side) and writes (3 each side) never overlap those of another edge (edges lie on the 8x8 grid), so all edges of a
// direction are independent; vertical edges of the whole picture are done before the horizontal ones (loopFilterPic :130-158).
#pragma once

struct DbkParams { int32_t sliceType, qp; int32_t refPoc[2][16]; };

HM_CONST uint8_t HM_DBK_TC[54] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };   // sm_tcTable :57
HM_CONST uint8_t HM_DBK_BETA[52] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };   // sm_betaTable :62
HM_CONST uint8_t HM_DBK_CHROMA_SCALE[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };

// boundary strength of the edge on the left of (dir 0) / above (dir 1) the 4x4 partition (x4, y4) (picture coordinates, on the 8x8
// grid); 0 when the edge is neither a transform nor a prediction edge.  xSetEdgefilterTU :269, xSetEdgefilterPU :293,
// xGetBoundaryStrengthSingle :411.
HM_DEV inline int dbk_bs(const Params *P, const FrameBuf *fb, const DbkParams *dp, int dir, int x4, int y4)
{
  const int a = (y4 >> 4) * P->wCtu + (x4 >> 4), z = hm_r2z((y4 & 15) * 16 + (x4 & 15));
  const CtuMeta *mq = fb->meta + a;
  if (mq->part[z] == SIZE_NONE) return 0;
  const int size4 = 16 >> mq->depth[z], rel = (dir == 0 ? x4 : y4) & (size4 - 1);
  int tus = size4 >> mq->tr[z]; if (tus < 1) tus = 1;
  const int tuEdge = (rel & (tus - 1)) == 0;
  int filt = tuEdge;                                                // CU edges are transform edges; picture borders never get here
  if (!filt) {
    const int ps = mq->part[z], h = size4 >> 1, q = size4 >> 2;
    if (dir == 0) filt = ((ps == SIZE_Nx2N || ps == SIZE_NxN) && rel == h) || (ps == SIZE_nLx2N && rel == q) || (ps == SIZE_nRx2N && rel == size4 - q);
    else filt = ((ps == SIZE_2NxN || ps == SIZE_NxN) && rel == h) || (ps == SIZE_2NxnU && rel == q) || (ps == SIZE_2NxnD && rel == size4 - q);
    if (!filt) return 0;
  }
  const int px4 = x4 - (dir == 0), py4 = y4 - (dir == 1);
  const int ap = (py4 >> 4) * P->wCtu + (px4 >> 4), zp = hm_r2z((py4 & 15) * 16 + (px4 & 15));
  const CtuMeta *mp = fb->meta + ap;
  if (mp->pred[zp] == MODE_INTRA || mq->pred[z] == MODE_INTRA) return 2;
  if (tuEdge && (((mq->cbf[0][z] >> mq->tr[z]) & 1) || ((mp->cbf[0][zp] >> mp->tr[zp]) & 1))) return 1;
  const InterMeta *iq = fb->imeta + a, *ip = fb->imeta + ap;
  int rp[2], rq[2]; MvD vp[2], vq[2];
  for (int l = 0; l < 2; l++) {                                       // pictures are compared by POC; "none" = INT_MIN with a zero MV
    const int riP = ip->refIdx[l][zp], riQ = iq->refIdx[l][z];
    rp[l] = riP < 0 ? (int)0x80000000 : dp->refPoc[l][riP]; rq[l] = riQ < 0 ? (int)0x80000000 : dp->refPoc[l][riQ];
    vp[l].x = vp[l].y = vq[l].x = vq[l].y = 0;
    if (riP >= 0) vp[l] = ip->mv[l][zp];
    if (riQ >= 0) vq[l] = iq->mv[l][z];
  }
#define HM_DBK_FAR(A, B) (hm_abs((A).x - (B).x) >= 4 || hm_abs((A).y - (B).y) >= 4)
  if (dp->sliceType == 0) {                                           // B slice
    if ((rp[0] == rq[0] && rp[1] == rq[1]) || (rp[0] == rq[1] && rp[1] == rq[0])) {
      if (rp[0] != rp[1]) {
        if (rp[0] == rq[0]) return (HM_DBK_FAR(vq[0], vp[0]) || HM_DBK_FAR(vq[1], vp[1])) ? 1 : 0;
        return (HM_DBK_FAR(vq[1], vp[0]) || HM_DBK_FAR(vq[0], vp[1])) ? 1 : 0;
      }
      return ((HM_DBK_FAR(vq[0], vp[0]) || HM_DBK_FAR(vq[1], vp[1])) && (HM_DBK_FAR(vq[1], vp[0]) || HM_DBK_FAR(vq[0], vp[1]))) ? 1 : 0;
    }
    return 1;
  }
  return (rp[0] != rq[0] || HM_DBK_FAR(vq[0], vp[0])) ? 1 : 0;
#undef HM_DBK_FAR
}

// QP of the edge between the partition (x4, y4) and its left (dir 0) / upper (dir 1) neighbour: (QP_P + QP_Q + 1) >> 1 of the two CUs
// (xEdgeFilterLuma :582-586).  Without cu_qp_delta every CU carries the slice QP.
HM_DEV inline int dbk_part_qp(const Params *P, const FrameBuf *fb, int x4, int y4)
{
  const int a = (y4 >> 4) * P->wCtu + (x4 >> 4), z = hm_r2z((y4 & 15) * 16 + (x4 & 15));
  const CtuDqp d = fb->dqp->out[a];
  return z < d.firstZ ? d.refQp : d.qp;
}
HM_DEV inline int dbk_edge_qp(const Params *P, const FrameBuf *fb, const DbkParams *dp, int dir, int x4, int y4)
{
  if (!fb->dqp) return dp->qp;
  return (dbk_part_qp(P, fb, x4 - (dir == 0), y4 - (dir == 1)) + dbk_part_qp(P, fb, x4, y4) + 1) >> 1;
}

// one luma edge segment: 4 lines of 8 samples across the edge; s[line][0..7] = p3 p2 p1 p0 | q0 q1 q2 q3 (xEdgeFilterLuma :540, xPelFilterLuma :800)
HM_DEV inline void dbk_filter_luma4(int s[4][8], int bs, int qp, int bd)
{
  const int scale = 1 << (bd - 8);
  const int tc = HM_DBK_TC[hm_clip3(0, 53, qp + 2 * (bs - 1))] * scale, beta = HM_DBK_BETA[hm_clip3(0, 51, qp)] * scale;
  const int sideThr = (beta + (beta >> 1)) >> 3, thrCut = tc * 10, maxv = (1 << bd) - 1;
  const int dp0 = hm_abs(s[0][1] - 2 * s[0][2] + s[0][3]), dq0 = hm_abs(s[0][4] - 2 * s[0][5] + s[0][6]);
  const int dp3 = hm_abs(s[3][1] - 2 * s[3][2] + s[3][3]), dq3 = hm_abs(s[3][4] - 2 * s[3][5] + s[3][6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, dpp = dp0 + dp3, dqq = dq0 + dq3;
  if (d0 + d3 >= beta) return;
  const int fP = dpp < sideThr, fQ = dqq < sideThr;
  const int sw0 = (hm_abs(s[0][0] - s[0][3]) + hm_abs(s[0][7] - s[0][4])) < (beta >> 3) && 2 * d0 < (beta >> 2) && hm_abs(s[0][3] - s[0][4]) < ((tc * 5 + 1) >> 1);
  const int sw3 = (hm_abs(s[3][0] - s[3][3]) + hm_abs(s[3][7] - s[3][4])) < (beta >> 3) && 2 * d3 < (beta >> 2) && hm_abs(s[3][3] - s[3][4]) < ((tc * 5 + 1) >> 1);
  const int sw = sw0 && sw3;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m0 = s[i][0], m1 = s[i][1], m2 = s[i][2], m3 = s[i][3], m4 = s[i][4], m5 = s[i][5], m6 = s[i][6], m7 = s[i][7];
    if (sw) {
      s[i][3] = hm_clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
      s[i][4] = hm_clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
      s[i][2] = hm_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
      s[i][5] = hm_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
      s[i][1] = hm_clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
      s[i][6] = hm_clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    } else {
      int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
      if (hm_abs(delta) < thrCut) {
        delta = hm_clip3(-tc, tc, delta);
        s[i][3] = hm_clip3(0, maxv, m3 + delta); s[i][4] = hm_clip3(0, maxv, m4 - delta);
        const int tc2 = tc >> 1;
        if (fP) s[i][2] = hm_clip3(0, maxv, m2 + hm_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
        if (fQ) s[i][5] = hm_clip3(0, maxv, m5 + hm_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
      }
    }
  }
}
HM_DEV inline int dbk_chroma_tc(int qp, int bs, int bd)
{ // xEdgeFilterChroma :752-769 (4:2:0, cb/cr_qp_offset 0)
  int q = qp;
  if (q >= 58) q -= 6; else if (q >= 0) q = HM_DBK_CHROMA_SCALE[q];
  return HM_DBK_TC[hm_clip3(0, 53, q + 2 * (bs - 1))] * (1 << (bd - 8));
}

#if !defined(HM355_HOSTSIM)
// mode 0: vertical luma edges, 1: vertical chroma edges, 2: horizontal luma edges, 3: horizontal chroma edges.
// grid: x = ceil(items along the row / 64), y = segment rows, z = picture of the batch.
extern "C" __global__ void __launch_bounds__(64) hm355_dbk_kernel(const Params *P, const DbkParams *dps, int mode)
{
  const FrameBuf *fb = P->frames + blockIdx.z; const DbkParams *dp = dps + blockIdx.z;
  const int w = P->width, h = P->height, bd = P->bitDepth;
  const int i = (int)(blockIdx.x * 64 + threadIdx.x), j = (int)blockIdx.y;
  if (mode == 0) {                                                    // lane i: edge x = 8*(i+1); j: 4-row segment
    const int x = 8 * (i + 1), y = 4 * j;
    if (x >= w || y >= h) return;
    const int bs = dbk_bs(P, fb, dp, 0, x >> 2, y >> 2);
    if (!bs) return;
    Pel *p = fb->rec[0] + (size_t)y * P->stride[0] + x - 4; const int st = P->stride[0];
    int s[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++) { const uint4 v = *(const uint4 *)(p + r * st); const Pel *q = (const Pel *)&v;
#pragma unroll
      for (int k = 0; k < 8; k++) s[r][k] = q[k]; }
    dbk_filter_luma4(s, bs, dbk_edge_qp(P, fb, dp, 0, x >> 2, y >> 2), bd);
#pragma unroll
    for (int r = 0; r < 4; r++) { uint4 v; Pel *q = (Pel *)&v;
#pragma unroll
      for (int k = 0; k < 8; k++) q[k] = (Pel)s[r][k];
      *(uint4 *)(p + r * st) = v; }
  } else if (mode == 2) {                                             // lane i: columns 4i..4i+3; j: edge y = 8*(j+1)
    const int x = 4 * i, y = 8 * (j + 1);
    if (x >= w || y >= h) return;
    const int bs = dbk_bs(P, fb, dp, 1, x >> 2, y >> 2);
    if (!bs) return;
    Pel *p = fb->rec[0] + (size_t)(y - 4) * P->stride[0] + x; const int st = P->stride[0];
    int s[4][8];
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint2 v = *(const uint2 *)(p + k * st); const Pel *q = (const Pel *)&v;
#pragma unroll
      for (int r = 0; r < 4; r++) s[r][k] = q[r]; }
    dbk_filter_luma4(s, bs, dbk_edge_qp(P, fb, dp, 1, x >> 2, y >> 2), bd);
#pragma unroll
    for (int k = 1; k < 7; k++) { uint2 v; Pel *q = (Pel *)&v;
#pragma unroll
      for (int r = 0; r < 4; r++) q[r] = (Pel)s[r][k];
      *(uint2 *)(p + k * st) = v; }
  } else if (mode == 1) {                                             // chroma, vertical: edge luma x = 16*(i+1); j: 4-luma-row partition = 2 chroma rows
    const int x = 16 * (i + 1), y = 4 * j;
    if (x >= w || y >= h) return;
    const int bs = dbk_bs(P, fb, dp, 0, x >> 2, y >> 2);
    if (bs <= 1) return;
    const int tc = dbk_chroma_tc(dbk_edge_qp(P, fb, dp, 0, x >> 2, y >> 2), bs, bd), maxv = (1 << bd) - 1;
    for (int c = 1; c < 3; c++) {
      const int st = P->stride[c]; Pel *p = fb->rec[c] + (size_t)(y >> 1) * st + (x >> 1);
#pragma unroll
      for (int r = 0; r < 2; r++) {                                    // xPelFilterChroma :870
        Pel *q = p + r * st; const int m2 = q[-2], m3 = q[-1], m4 = q[0], m5 = q[1];
        const int delta = hm_clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        q[-1] = (Pel)hm_clip3(0, maxv, m3 + delta); q[0] = (Pel)hm_clip3(0, maxv, m4 - delta);
      }
    }
  } else {                                                            // chroma, horizontal: lane i: 4-luma-column partition = 2 chroma columns; j: edge luma y = 16*(j+1)
    const int x = 4 * i, y = 16 * (j + 1);
    if (x >= w || y >= h) return;
    const int bs = dbk_bs(P, fb, dp, 1, x >> 2, y >> 2);
    if (bs <= 1) return;
    const int tc = dbk_chroma_tc(dbk_edge_qp(P, fb, dp, 1, x >> 2, y >> 2), bs, bd), maxv = (1 << bd) - 1;
    for (int c = 1; c < 3; c++) {
      const int st = P->stride[c]; Pel *p = fb->rec[c] + (size_t)(y >> 1) * st + (x >> 1);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        Pel *q = p + r; const int m2 = q[-2 * st], m3 = q[-st], m4 = q[0], m5 = q[st];
        const int delta = hm_clip3(-tc, tc, ((((m4 - m3) << 2) + m2 - m5 + 4) >> 3));
        q[-st] = (Pel)hm_clip3(0, maxv, m3 + delta); q[0] = (Pel)hm_clip3(0, maxv, m4 - delta);
      }
    }
  }
}

// ---- TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64-139), layer 0, integer part: per CTU the sum and the sum of squares of the original
// luma samples of its four quadrants (split at half of the unit's size inside the picture).  One wavefront per CTU, lanes along the rows
// (coalesced 128-byte reads), 64-bit accumulators, butterfly reduction.  The caller derives activity = 1 + min "variance" and the QP offsets
// in double precision exactly as the reference does (sums[a][0..3] = sum, [4..7] = sum of squares of quadrant TL, TR, BL, BR).
extern "C" __global__ void __launch_bounds__(64) hm355_preanalyze_kernel(const Params *P, int frame, unsigned long long *sums)
{
  const FrameBuf *fb = P->frames + frame;
  const int a = (int)blockIdx.x, cx = a % P->wCtu, cy = a / P->wCtu, lane = (int)threadIdx.x;
  const int x0 = cx * 64, y0 = cy * 64;
  const int w = P->width - x0 < 64 ? P->width - x0 : 64, h = P->height - y0 < 64 ? P->height - y0 : 64;
  unsigned long long s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < w) {
    const int right = lane >= (w >> 1);
    const Pel *p = fb->org[0] + (size_t)y0 * P->stride[0] + x0 + lane;
    for (int y = 0; y < h; y++) {
      const int v = p[(size_t)y * P->stride[0]], k = (y >= (h >> 1) ? 2 : 0) + right;
      s[k] += (unsigned long long)v; s[4 + k] += (unsigned long long)(v * v);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    unsigned long long v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) sums[(size_t)a * 8 + k] = v;
  }
}

// ---- device-resident reference pictures: what later pictures read of a finished one ----
// mode 0..2: border extension of plane c (TComPicYuv::extendPicBorder, TComPicYuv.cpp:171) from the slot's reconstruction into a
//            plane with HM_REF_MARGIN (>> 1 for chroma) samples on every side; one lane per destination sample, rows contiguous.
// mode 3:    TComPic::compressMotion (TComDataCU::compressMV :3328, TComCUMvField::compress): every 16x16 block takes the prediction
//            mode, MVs and reference indices of its first 4x4 partition; one lane per partition.
extern "C" __global__ void __launch_bounds__(64) hm355_ref_kernel(const Params *P, int frame, int mode, Pel *dst, uint8_t *predMode, MvD *mv0, MvD *mv1, int8_t *ri0, int8_t *ri1)
{
  const FrameBuf *fb = P->frames + frame;
  if (mode < 3) {
    const int c = mode, cw = P->width >> (c ? 1 : 0), ch = P->height >> (c ? 1 : 0), mg = HM_REF_MARGIN >> (c ? 1 : 0), st = cw + 2 * mg;
    const int x = (int)(blockIdx.x * 64 + threadIdx.x), y = (int)blockIdx.y;
    if (x >= st) return;
    const int sx = hm_clip3(0, cw - 1, x - mg), sy = hm_clip3(0, ch - 1, y - mg);
    dst[(size_t)y * st + x] = fb->rec[c][(size_t)sy * P->stride[c] + sx];
  } else {
    const int i = (int)(blockIdx.x * 64 + threadIdx.x), n = P->wCtu * P->hCtu * 256;
    if (i >= n) return;
    const int a = i >> 8, z = i & 255, z0 = z & ~15;
    predMode[i] = fb->meta[a].pred[z0];
    MvD zero; zero.x = zero.y = 0;
    if (fb->imeta) { const InterMeta *m = fb->imeta + a; mv0[i] = m->mv[0][z0]; mv1[i] = m->mv[1][z0]; ri0[i] = m->refIdx[0][z0]; ri1[i] = m->refIdx[1][z0]; }
    else { mv0[i] = zero; mv1[i] = zero; ri0[i] = -1; ri1[i] = -1; }
  }
}
#endif
