// hm355 -- candidates in lanes: independent 4x4 transform blocks evaluated one per lane.
//
// The reference tries the RD candidates of a prediction unit one after the other, each from the same CABAC snapshot
// (TEncSearch::estIntraPredQT :2473-2490, the transform-skip trial of xRecurIntraCodingQT :1452-1530, the five chroma
// modes of estIntraPredChromaQT :2740-2800).  None of them reads what another one wrote, so for 4x4 blocks -- where one
// candidate is far too little work for 64 lanes and where most of the evaluations of a CTU are -- the wavefront takes a
// whole candidate list at once: lane k owns candidate k from the prediction to the bit count (prediction, residual,
// DST/DCT or transform skip, RDOQ, de-quantisation, inverse transform, reconstruction, SSE, CABAC bit estimate), in the
// reference's arithmetic and operation order.  The decision among the candidates afterwards is the reference's
// sequential comparison (strict "<", first candidate wins ties) on the per-lane costs.
//
// Per-lane state lives in registers (the 16 samples / coefficients of the block) and in small lane-indexed LDS arrays
// that overlay the transform buffers, which are idle meanwhile.  Included from hm355_core.h.
#pragma once

#ifdef HM355_HOSTSIM
#define HM_FINL static inline
#else
#define HM_FINL __device__ __forceinline__
#endif
#define HM_SL 22                          // jobs (lanes in use) per batch: 11 luma candidates x {transform, transform skip}
// compact numbering of the contexts a 4x4 block's syntax touches (per-lane copies in Simt4A::ctx)
enum { S4_SIG = 0, S4_ONE = 9, S4_ABS = 13, S4_LX = 14, S4_LY = 17, S4_CBF = 20, S4_TSKIP = 21, S4_MODE = 22, S4_NCTX = 24 };
// bit costs of the batch's start state (Simt4A::tab)
enum { T4_SIG = 0, T4_ONE = 18, T4_ABS = 26, T4_LASTX = 28, T4_LASTY = 32, T4_CBF = 36, T4_N = 40 };
struct Simt4A {                           // overlays Shared::bufA
  double cost[16][HM_SL];                 // RDOQ: cost of the level decided at each scan position
  double outCost[HM_SL];
  uint32_t outDist[HM_SL], outBits[HM_SL];
  int32_t tab[T4_N];
  uint8_t ctx[S4_NCTX][HM_SL];            // context states of each job
  uint8_t outCbf[HM_SL], pad[2];
  uint8_t lps[128];                       // LPS transitions
};
struct Simt4B {                           // overlays Shared::u behind the reference sample lines (RefLds::refMain onwards)
  int32_t cs[16][HM_SL];                  // coefficients in scan order
  int32_t dc[16][HM_SL];                  // level at decision time (low half) | final signed level (high half), scan order
  uint8_t scan[3][16];                    // scan position -> raster position of the three 4x4 scans
  uint8_t sigIdx[3][16];                  // scan position -> significance context increment
};
static_assert(sizeof(Simt4A) <= sizeof(((Shared *)0)->bufA), "Simt4A overlays bufA");
static_assert(offsetof(RefLds, refMain) + sizeof(Simt4B) <= sizeof(((Shared *)0)->u), "Simt4B overlays the tail of the LDS union");
static_assert(offsetof(RefLds, refMain) % 8 == 0, "alignment of the overlay");
HM_DEV inline Simt4A *s4_a(Shared *e) { return (Simt4A *)e->bufA; }
HM_DEV inline Simt4B *s4_b(Shared *e) { return (Simt4B *)((char *)&e->u + offsetof(RefLds, refMain)); }

// the three 4x4 coefficient scans (TComRom.cpp:140-225; checked against the generated tables by s4_setup's callers' tests) and their inverses
HM_DEV constexpr int s4_scan(int type, int i)
{
  constexpr uint8_t t[3][16] = { {0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
                                 {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15} };
  return t[type][i];
}
HM_DEV constexpr int s4_inv_scan(int type, int r)
{
  constexpr uint8_t t[3][16] = { {0, 2, 5, 9, 1, 4, 8, 12, 3, 7, 11, 14, 6, 10, 13, 15}, {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
                                 {0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15} };
  return t[type][r];
}
struct S4Par {                            // wave-uniform parameters of a batch
  int chroma, bitDepth, qBits, quantCoef, tshift; double errScale, lambda; int64_t rdFactor;
  int dqShift, dqScale, dqMin, dqMax;
};
HM_DEV inline S4Par s4_params(const Shared *e, int chroma)
{
  S4Par p;
  p.chroma = chroma; p.bitDepth = e->bitDepth; p.tshift = 15 - e->bitDepth - 2;
  p.qBits = 14 + e->fb.qpPer[chroma] + p.tshift; p.quantCoef = HM_QUANT_SCALES[e->fb.qpRem[chroma]];
  p.errScale = e->fb.errScale[chroma][0]; p.lambda = chroma ? e->fb.lambdaC : e->fb.lambda; p.rdFactor = e->fb.rdFactor[chroma];
  p.dqShift = 6 - (p.tshift + e->fb.qpPer[chroma]); p.dqScale = HM_INV_QUANT_SCALES[e->fb.qpRem[chroma]];
  int tgt = 25 + p.dqShift; if (tgt > 16) tgt = 16;
  p.dqMin = -(1 << (tgt - 1)); p.dqMax = (1 << (tgt - 1)) - 1;
  return p;
}

// Tables of a batch: bit costs of the start state `cb` for one component type (estBit, TEncSbac.cpp:1717-1956), the
// per-job context copies, the scans.  cbfCtx: index inside C_QT_CBF of the cbf RDOQ prices; modeCtx: the context of the
// prediction-mode bin the jobs code (C_INTRA_LUMA / C_CHROMA_PRED).
HM_DEV inline void s4_setup(Shared *e, const Cabac *cb, int chroma, int cbfCtx, int cbfCodeCtx, int modeCtx, int jobs)
{
  Simt4A *A = s4_a(e); Simt4B *B = s4_b(e);
  const int sigOff = C_SIG + (chroma ? 28 : 0), oneOff = C_ONE + (chroma ? 16 : 0), absOff = C_ABS + (chroma ? 4 : 0);
  const int lxOff = C_LASTX + (chroma ? 15 : 0), lyOff = C_LASTY + (chroma ? 15 : 0);
  HM_PAR_FOR(i, T4_N) {
    int v = 0;
    if (i < T4_ONE) v = HM_LT()->ebits[cb->s[sigOff + (i >> 1)] ^ (i & 1)];
    else if (i < T4_ABS) v = HM_LT()->ebits[cb->s[oneOff + ((i - T4_ONE) >> 1)] ^ (i & 1)];
    else if (i < T4_LASTX) v = HM_LT()->ebits[cb->s[absOff] ^ (i & 1)];
    else if (i < T4_CBF) { // cost of the last-position group index g = 0..3: g ones, then a zero unless g is the maximum (xGetRateLast, TComTrQuant.cpp:2815)
      const int g = (i - T4_LASTX) & 3, off = i < T4_LASTY ? lxOff : lyOff;
      for (int c = 0; c < g; c++) v += HM_LT()->ebits[cb->s[off + c] ^ 1];
      if (g < 3) v += HM_LT()->ebits[cb->s[off + g] ^ 0];
    } else if (i < T4_CBF + 2) v = HM_LT()->ebits[cb->s[C_QT_CBF + cbfCtx] ^ (i & 1)];
    A->tab[i] = v;
  }
  HM_PAR_FOR(i, 128) A->lps[i] = HM_NEXT_LPS[i];
  HM_PAR_FOR(i, S4_NCTX * HM_SL) {
    const int j = i / HM_SL, k = i - j * HM_SL;
    int c;
    if (j < S4_ONE) c = sigOff + j; else if (j < S4_ABS) c = oneOff + (j - S4_ONE); else if (j == S4_ABS) c = absOff;
    else if (j < S4_LY) c = lxOff + (j - S4_LX); else if (j < S4_CBF) c = lyOff + (j - S4_LY);
    else if (j == S4_CBF) c = C_QT_CBF + cbfCodeCtx; else if (j == S4_TSKIP) c = C_TSKIP + (chroma ? 1 : 0); else c = modeCtx;
    if (k < jobs) A->ctx[j][k] = cb->s[c];
  }
  HM_PAR_FOR(i, 48) {
    const int ty = i >> 4, sp = i & 15, blk = e->tab->scan[ty][0][sp];
    B->scan[ty][sp] = (uint8_t)blk; B->sigIdx[ty][sp] = HM_CTX_IND_MAP_4x4[blk];
#ifdef HM355_HOSTSIM
    if (blk != s4_scan(ty, sp) || s4_inv_scan(ty, blk) != sp) abort();   // the constexpr scans must equal the generated tables
#endif
  }
  HM_SYNC();
}

// one context-coded bin on job k's private context copy; frac counts Q15 bits
HM_DEV inline void s4_bin(const Shared *e, Simt4A *A, int k, uint32_t *frac, int c, int bin)
{
  const int st = A->ctx[c][k];
  *frac += (uint32_t)HM_LT()->ebits[st ^ bin];
  A->ctx[c][k] = (uint8_t)(bin == (st & 1) ? (st < 124 ? st + 2 : st) : A->lps[st]);
}

// one predicted sample of a 4x4 block of component type `chroma` (TComPrediction::predIntraAng, TComPrediction.cpp:182-840;
// the arithmetic of pred_intra), from the unfiltered reference lines in slot `rs` of RefLds
HM_FINL int s4_pred_sample(const Shared *e, int rs, int chroma, int mode, int x, int y, int dcVal, int bitDepth)
{
  const Pel *top = e->u.ref.refTop[rs], *left = e->u.ref.refLeft[rs];
  if (mode == PLANAR_IDX) {
    const int hor = (left[y + 1] << 2) + 4 + (x + 1) * (top[5] - left[y + 1]);
    const int ver = (top[x + 1] << 2) + (y + 1) * (left[5] - top[x + 1]);
    return (hor + ver) >> 3;
  }
  if (mode == DC_IDX) {
    if (!chroma) {
      if (x == 0 && y == 0) return (top[1] + left[1] + 2 * dcVal + 2) >> 2;
      if (y == 0) return (top[x + 1] + 3 * dcVal + 2) >> 2;
      if (x == 0) return (left[y + 1] + 3 * dcVal + 2) >> 2;
    }
    return dcVal;
  }
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER_IDX : -(mode - HOR_IDX);
  const int absAng = HM_ANG_TABLE[hm_abs(angMode)], invAngle = HM_INV_ANG_TABLE[hm_abs(angMode)];
  const int angle = angMode < 0 ? -absAng : absAng;
  const Pel *mainR = isVer ? top : left, *sideR = isVer ? left : top;
  const int xx = isVer ? x : y, yy = isVer ? y : x;
  if (angle == 0) {
    int v = mainR[xx + 1];
    if (!chroma && xx == 0) v = hm_clip3(0, (1 << bitDepth) - 1, v + ((sideR[yy + 1] - sideR[0]) >> 1));
    return v;
  }
  const int deltaPos = (yy + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
  const int i0 = xx + di + 1;
  const int a = i0 >= 0 ? mainR[i0] : sideR[(128 - i0 * invAngle) >> 8];
  if (!df) return a;
  const int i1 = i0 + 1;
  const int b = i1 >= 0 ? mainR[i1] : sideR[(128 - i1 * invAngle) >> 8];
  return ((32 - df) * a + df * b + 16) >> 5;
}

// 4-point core transforms of one block held by one lane (xTrMxN / xITrMxN, TComTrQuant.cpp:836-935; DST for intra luma)
HM_FINL int s4_tm(int dst, int k, int j)
{
  if (dst) return HM_DST4[k * 4 + j];
  const int m = (k * 8 * (2 * j + 1)) & 127;      // the row k * 8 of the 32-point matrix (load_tmat)
  if (m <= 32) return HM_DCT_C[m]; if (m <= 64) return -HM_DCT_C[64 - m]; if (m <= 96) return -HM_DCT_C[m - 64]; return HM_DCT_C[128 - m];
}
HM_FINL void s4_fwd(int32_t *blk, int dst, int bitDepth)
{
  const int s1 = 2 + bitDepth + 6 - 15, a1 = 1 << (s1 - 1), s2 = 8, a2 = 128;
  int32_t t[16];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 4; k++) { int32_t acc = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) acc += s4_tm(dst, k, i) * blk[j * 4 + i];
      t[k * 4 + j] = (acc + a1) >> s1; }
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int k = 0; k < 4; k++) { int32_t acc = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) acc += s4_tm(dst, k, i) * t[j * 4 + i];
      blk[k * 4 + j] = (acc + a2) >> s2; }
}
HM_FINL void s4_inv(int32_t *blk, int dst, int bitDepth)
{
  const int s1 = 7, s2 = 20 - bitDepth;
  int32_t t[16];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) { int32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) acc += s4_tm(dst, k, i) * blk[k * 4 + j];
      t[j * 4 + i] = hm_clip3(-32768, 32767, (acc + (1 << (s1 - 1))) >> s1); }
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) { int32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) acc += s4_tm(dst, k, i) * t[k * 4 + j];
      blk[j * 4 + i] = hm_clip3(-32768, 32767, (acc + (1 << (s2 - 1))) >> s2); }
}

// xGetICRate, TComTrQuant.cpp:2725-2800, on the batch's bit-cost table (one context set: a 4x4 block has one coefficient group)
HM_FINL int s4_ic_rate(const int32_t *tab, uint32_t absLevel, int c1, int goRice, int c1Idx, int c2Idx)
{
  int rate = 32768;
  const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < (3u << goRice)) { length = symbol >> goRice; rate += (int)((length + 1 + goRice) << 15); }
    else {
      length = (uint32_t)goRice; symbol -= (3u << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((3 + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < 8) { rate += tab[T4_ONE + c1 * 2 + 1]; if (c2Idx < 1) rate += tab[T4_ABS + 1]; }
  } else if (absLevel == 1) rate += tab[T4_ONE + c1 * 2];
  else if (absLevel == 2) { rate += tab[T4_ONE + c1 * 2 + 1]; rate += tab[T4_ABS]; }
  else rate = 0;
  return rate;
}

// RDOQ of job k's 4x4 block (TComTrQuant::xRateDistOptQuant, TComTrQuant.cpp:1974-2511, one coefficient group): coefficients in
// B->cs[.][k] (scan order); leaves the signed levels in the high halves of B->dc[.][k] and returns the sum of their magnitudes.
HM_DEV inline int s4_rdoq(Simt4A *A, Simt4B *B, const S4Par &p, int k, int scanType)
{
  const int32_t *tab = A->tab;
  const int qBits = p.qBits, half = 1 << (qBits - 1);
  const int64_t cap = 2147483647LL - (1LL << (qBits - 1));
  const double lambda = p.lambda, errScale = p.errScale;
  double blockUncoded = 0, baseCost = 0;
  int last = -1, c1 = 1, c2 = 0, c1Idx = 0, c2Idx = 0, goRice = 0;
  for (int i = 15; i >= 0; i--) {
    const int32_t sc = B->cs[i][k];
    const int64_t tl = (int64_t)hm_abs(sc) * p.quantCoef;
    const int32_t lvlD = (int32_t)(tl < cap ? tl : cap);
    uint32_t mx = (uint32_t)((lvlD + half) >> qBits); if (mx > 32767u) mx = 32767u;
    const double err = (double)lvlD, c0 = err * err * errScale;
    blockUncoded += c0;
    if (mx > 0 && last < 0) last = i;
    uint32_t level = 0;
    if (last >= 0) {
      const int isLast = (i == last), si = B->sigIdx[scanType][i];
      double cCoeff;
      if (!isLast && mx < 3) { const double s0 = lambda * (double)tab[T4_SIG + si * 2]; cCoeff = c0 + s0; }   // xGetCodedLevel :2660
      else cCoeff = HM_MAX_DOUBLE;
      if (mx > 0) {
        double currCostSig = 0;
        if (!isLast) currCostSig = lambda * (double)tab[T4_SIG + si * 2 + 1];
        const uint32_t minAbs = mx > 1 ? mx - 1 : 1;
        for (int al = (int)mx; al >= (int)minAbs; al--) {
          const double de = (double)(lvlD - (int32_t)((uint32_t)al << qBits));
          const double dist = de * de * errScale;
          const double rc = lambda * (double)s4_ic_rate(tab, (uint32_t)al, c1, goRice, c1Idx, c2Idx);
          double cc = dist + rc;
          cc += currCostSig;
          if (cc < cCoeff) { level = (uint32_t)al; cCoeff = cc; }
        }
      }
      A->cost[i][k] = cCoeff;
      baseCost += cCoeff;
      const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
      if (level >= baseLevel && level > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
      if (level >= 1) c1Idx++;
      if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
      else if (c1 < 3 && c1 > 0 && level) c1++;
    } else baseCost += c0;
    B->dc[i][k] = (int32_t)level;
  }
  if (last < 0) return 0;
  double bestCost = blockUncoded + lambda * (double)tab[T4_CBF];            // TComTrQuant.cpp:2310-2316
  baseCost += lambda * (double)tab[T4_CBF + 1];
  int bestLastP1 = 0;
  for (int i = last; i >= 0; i--) {
    const int lev = B->dc[i][k], si = B->sigIdx[scanType][i];
    if (lev) {
      const int blkPos = B->scan[scanType][i];
      int py = blkPos >> 2, px = blkPos & 3;
      if (scanType == SCAN_VER) { const int t = px; px = py; py = t; }
      const double costLast = lambda * (double)(tab[T4_LASTX + px] + tab[T4_LASTY + py]);
      const double cSig = (i == last) ? 0.0 : lambda * (double)tab[T4_SIG + si * 2 + 1];
      const double t1 = baseCost + costLast;
      const double totalCost = t1 - cSig;
      if (totalCost < bestCost) { bestLastP1 = i + 1; bestCost = totalCost; }
      if (lev > 1) break;
      const int64_t tl = (int64_t)hm_abs(B->cs[i][k]) * p.quantCoef;
      const double err = (double)(int32_t)(tl < cap ? tl : cap);
      baseCost -= A->cost[i][k]; baseCost += err * err * errScale;
    } else baseCost -= lambda * (double)tab[T4_SIG + si * 2];
  }
  // levels with signs, truncated at the chosen last position
  int absSum = 0, lastNZ = -1, firstNZ = 16, parity = 0;
  for (int i = 0; i < 16; i++) {
    const int dec = B->dc[i][k], lv = i < bestLastP1 ? dec : 0;
    absSum += lv; parity ^= lv & 1;
    if (lv) { lastNZ = i; if (firstNZ == 16) firstNZ = i; }
    B->dc[i][k] = dec | (int32_t)((uint32_t)(B->cs[i][k] < 0 ? -lv : lv) << 16);
  }
  // sign bit hiding, TComTrQuant.cpp:2380-2510 (the block's only coefficient group is its last one)
  if (absSum >= 2 && lastNZ - firstNZ >= 4) {
    const uint32_t signbit = (B->dc[firstNZ][k] >> 16) > 0 ? 0 : 1;
    if (signbit != (uint32_t)parity) {
      const int64_t I64MAX = 0x7fffffffffffffffLL;
      int64_t minCostInc = I64MAX, curCost = I64MAX; int minK = -1, finalChange = 0, curChange = 0;
      int wC1 = 1, wC1Idx = 0, wC2Idx = 0, wGoR = 0;          // the decision chain's state, walked again
      for (int q = last; q >= 0; --q) {
        const int v = B->dc[q][k]; const uint32_t dec = (uint32_t)(v & 0xffff); const int dv = v >> 16;
        const int ctxC1 = wC1, goR = wGoR, c1I = wC1Idx, c2I = wC2Idx;
        {
          const uint32_t baseLevel = (wC1Idx < 8) ? (2 + (wC2Idx < 1)) : 1;
          if (dec >= baseLevel && dec > (3u << wGoR)) wGoR = wGoR + 1 < 4 ? wGoR + 1 : 4;
          if (dec >= 1) wC1Idx++;
          if (dec > 1) { wC1 = 0; wC2Idx++; }
          else if (wC1 < 3 && wC1 > 0 && dec) wC1++;
        }
        if (q > lastNZ) continue;
        const int sc = B->cs[q][k];
        const int64_t tl = (int64_t)hm_abs(sc) * p.quantCoef;
        const int32_t lvlD = (int32_t)(tl < cap ? tl : cap);
        const int32_t deltaU = (int32_t)((lvlD - (int32_t)(dec << qBits)) >> (qBits - 8));
        const int si = B->sigIdx[scanType][q];
        const int sigRateDelta = (q == last) ? 0 : tab[T4_SIG + si * 2 + 1] - tab[T4_SIG + si * 2];
        int rateIncUp, rateIncDown = 0;
        if (dec > 0) {
          const int rateNow = s4_ic_rate(tab, dec, ctxC1, goR, c1I, c2I);
          rateIncUp = s4_ic_rate(tab, dec + 1, ctxC1, goR, c1I, c2I) - rateNow;
          rateIncDown = s4_ic_rate(tab, dec - 1, ctxC1, goR, c1I, c2I) - rateNow;
        } else rateIncUp = tab[T4_ONE + ctxC1 * 2];
        if (dv != 0) {
          const int64_t costUp = p.rdFactor * (-deltaU) + rateIncUp;
          int64_t costDown = p.rdFactor * (deltaU) + rateIncDown - ((hm_abs(dv) == 1) ? sigRateDelta : 0);
          if (lastNZ == q && hm_abs(dv) == 1) costDown -= (4 << 15);
          if (costUp < costDown) { curCost = costUp; curChange = 1; }
          else { curChange = -1; if (q == firstNZ && hm_abs(dv) == 1) curCost = I64MAX; else curCost = costDown; }
        } else {
          curCost = p.rdFactor * (-(hm_abs(deltaU))) + (1 << 15) + rateIncUp + sigRateDelta;
          curChange = 1;
          if (q < firstNZ) { const uint32_t thissign = sc < 0 ? 1u : 0u; if (thissign != signbit) curCost = I64MAX; }
        }
        if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minK = q; }
      }
      if (minK >= 0) {
        const int v = B->dc[minK][k]; int mv = v >> 16;
        if (mv == 32767 || mv == -32768) finalChange = -1;
        mv = (B->cs[minK][k] < 0) ? mv - finalChange : mv + finalChange;
        B->dc[minK][k] = (v & 0xffff) | (int32_t)((uint32_t)mv << 16);
      }
    }
  }
  return absSum;
}

// TEncSbac::codeCoeffNxN, TEncSbac.cpp:1172-1525, of a 4x4 block on job k's private contexts (bits only): the levels are the
// high halves of B->dc[.][src] (job `src` evaluated the block; k codes it)
HM_DEV inline void s4_code_coeff(const Shared *e, Simt4A *A, const Simt4B *B, int k, int src, int scanType, int tskipFlag, uint32_t *frac)
{
  s4_bin(e, A, k, frac, S4_TSKIP, tskipFlag);                      // codeTransformSkipFlags :988
  int last = -1;
  for (int i = 15; i >= 0; i--) if ((B->dc[i][src] >> 16) != 0) { last = i; break; }
  { // codeLastSignificantXY :1106 (group index = coordinate for a 4x4 block; no suffix bits)
    const int blkPos = B->scan[scanType][last];
    int py = blkPos >> 2, px = blkPos & 3;
    if (scanType == SCAN_VER) { const int t = px; px = py; py = t; }
    int q;
    for (q = 0; q < px; q++) s4_bin(e, A, k, frac, S4_LX + q, 1);
    if (px < 3) s4_bin(e, A, k, frac, S4_LX + q, 0);
    for (q = 0; q < py; q++) s4_bin(e, A, k, frac, S4_LY + q, 1);
    if (py < 3) s4_bin(e, A, k, frac, S4_LY + q, 0);
  }
  int numNonZero = 1, firstNZ = last;
  for (int i = last - 1; i >= 0; i--) {                             // significance flags; the last one is implied
    const int sig = (B->dc[i][src] >> 16) != 0;
    s4_bin(e, A, k, frac, S4_SIG + B->sigIdx[scanType][i], sig);
    if (sig) { numNonZero++; firstNZ = i; }
  }
  const int signHidden = (last - firstNZ >= 4);
  int c1 = 1, firstC2 = -1, escape = 0, idx = 0;
  for (int i = last; i >= 0 && idx < 8; i--) {
    const int a = hm_abs(B->dc[i][src] >> 16);
    if (!a) continue;
    const int sym = a > 1;
    s4_bin(e, A, k, frac, S4_ONE + c1, sym);
    if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = i; else escape = 1; }
    else if (c1 < 3 && c1 > 0) c1++;
    idx++;
  }
  if (c1 == 0 && firstC2 != -1) { const int sym = hm_abs(B->dc[firstC2][src] >> 16) > 2; s4_bin(e, A, k, frac, S4_ABS, sym); if (sym) escape = 1; }
  escape = escape || (numNonZero > 8);
  *frac += 32768u * (uint32_t)(signHidden ? numNonZero - 1 : numNonZero);
  if (escape) {
    int firstCoeff2 = 1; uint32_t goRice = 0; idx = 0;
    for (int i = last; i >= 0; i--) {
      const int a = hm_abs(B->dc[i][src] >> 16);
      if (!a) continue;
      const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
      if (a >= baseLevel) {                                         // xWriteCoefRemainExGolomb :337
        uint32_t sym = (uint32_t)(a - baseLevel);
        if (sym < (3u << goRice)) *frac += 32768u * ((sym >> goRice) + 1 + goRice);
        else { uint32_t len = goRice; sym -= (3u << goRice); while (sym >= (1u << len)) sym -= (1u << (len++)); *frac += 32768u * (3 + len + 1 - goRice + len); }
        if ((uint32_t)a > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
      }
      if (a >= 2) firstCoeff2 = 0;
      idx++;
    }
  }
}

// lane arrays: a small array per lane (registers on the device; the host twin keeps one per emulated lane)
#ifdef HM355_HOSTSIM
#define HM_LVA(T, name, n) T name[64][n]
#define HM_LVAK(name, k) name[k]
#else
#define HM_LVA(T, name, n) T name[n]
#define HM_LVAK(name, k) name
#endif

// Prediction, residual, transform (or transform skip), RDOQ, reconstruction and distortion of job k's 4x4 block.
//   rs: reference line slot; org/ps: original block; mode: prediction mode; ts: transform skip; scanType: coefficient scan
// Leaves the reconstruction in rec[16], the levels (raster order) in lv[16]; returns the distortion, *cbf = coded block flag.
HM_FINL uint32_t s4_eval(Shared *e, Simt4A *A, Simt4B *B, const S4Par &p, int k, int rs, const Pel *org, int ps, int mode, int ts, int scanType,
                               int dcVal, int32_t *rec, int32_t *lv, int *cbf)
{
  int32_t pred[16], blk[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    pred[i] = s4_pred_sample(e, rs, p.chroma, mode, i & 3, i >> 2, dcVal, p.bitDepth);
    const int r = org[(i >> 2) * ps + (i & 3)] - pred[i];
    blk[i] = ts ? (r << p.tshift) : r;                              // xTransformSkip, TComTrQuant.cpp:1874
  }
  if (!ts) s4_fwd(blk, !p.chroma, p.bitDepth);
#pragma unroll
  for (int i = 0; i < 16; i++)                                      // into scan order (the lane's own scan)
    B->cs[i][k] = scanType == SCAN_HOR ? blk[s4_scan(SCAN_HOR, i)] : (scanType == SCAN_VER ? blk[s4_scan(SCAN_VER, i)] : blk[s4_scan(SCAN_DIAG, i)]);
  const int absSum = s4_rdoq(A, B, p, k, scanType);
  *cbf = absSum > 0;
#pragma unroll
  for (int i = 0; i < 16; i++) lv[i] = 0;
  if (absSum > 0) {
    int32_t ls[16];
#pragma unroll
    for (int i = 0; i < 16; i++) ls[i] = B->dc[i][k] >> 16;
#pragma unroll
    for (int j = 0; j < 16; j++)                                    // back to raster order
      lv[j] = scanType == SCAN_HOR ? ls[s4_inv_scan(SCAN_HOR, j)] : (scanType == SCAN_VER ? ls[s4_inv_scan(SCAN_VER, j)] : ls[s4_inv_scan(SCAN_DIAG, j)]);
#pragma unroll
    for (int i = 0; i < 16; i++) {                                  // xDeQuant (flat), TComTrQuant.cpp:1276-1312
      const int c = hm_clip3(p.dqMin, p.dqMax, lv[i]);
      int v;
      if (p.dqShift > 0) v = (c * p.dqScale + (1 << (p.dqShift - 1))) >> p.dqShift;
      else v = (int)((unsigned)(c * p.dqScale) << (-p.dqShift));
      blk[i] = hm_clip3(-32768, 32767, v);
    }
    if (!ts) s4_inv(blk, !p.chroma, p.bitDepth);
    else {
      const int off = p.tshift == 0 ? 0 : (1 << (p.tshift - 1));
#pragma unroll
      for (int i = 0; i < 16; i++) blk[i] = (int16_t)((blk[i] + off) >> p.tshift);      // xITransformSkip, :1920 (stored as Pel)
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; i++) blk[i] = 0;
  }
  const int maxv = (1 << p.bitDepth) - 1, shiftSse = (p.bitDepth - 8) << 1;
  uint32_t sse = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int r = hm_clip3(0, maxv, pred[i] + (int16_t)blk[i]);
    rec[i] = r;
    const int d = org[(i >> 2) * ps + (i & 3)] - r; sse += (uint32_t)((d * d) >> shiftSse);
  }
  return sse;
}

// ------------------------------------------------------------------------------------------------
// luma: the RD candidates of one 4x4 prediction unit of an NxN CU (the candidate loop of estIntraPredQT :2473-2560 with the
// transform-skip trial of xRecurIntraCodingQT inside), all at once.  e->cur holds the CU's entry snapshot (CI_CURR_BEST);
// e->u.ref the PU's reference samples; e->rdModeList[0..numModes) the candidates; e->mpmPreds the PU's most probable modes.
// Writes the winner's decision (tr / cbf / ts / dirL of the partition, coefficients, reconstruction into ws->reco) and
// returns its mode / distortion / cost in e->outBits / e->outDistY / e->outRdCost.
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE void simt4_luma_pu(Shared *e, TU tv, int numModes)
{
  HM_ENTRY(e); numModes = HM_UNI(numModes); tv = hm_uni_struct(tv);
  const TU *t = &tv; CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const int z = t->cuZ + t->relZ, ps = e->stride[0], jobs = 2 * numModes;
  Simt4A *A = s4_a(e); Simt4B *B = s4_b(e);
  // bits every candidate spends before its own bins (xEncIntraHeader :965-1032 for the first PU of the CU): same contexts, same
  // values for all of them; none of those contexts is touched again inside the block
  uint32_t commonFrac;
  {
    CabacR r; cabr_load(e, r, &e->cur);
    r.frac &= 32767;
    if (t->relZ == 0) {
      if (e->im) { code_skip_flag(e, &r, t->cuZ); enc_bin(e, &r, C_PRED_MODE, 1); }
      if (t->cuDepth == 3) enc_bin(e, &r, C_PART, 0);
    }
    commonFrac = (uint32_t)r.frac;
  }
  s4_setup(e, &e->cur, 0, 0, 0, C_INTRA_LUMA, jobs);               // luma cbf of a TU below the CU root: context 0 (code_qt_cbf)
  const S4Par p = s4_params(e, 0);
  const Pel *org = e->fb.org[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x;
  int dcVal;
  { int s = 0; HM_PAR_FOR(i, 4) s += e->u.ref.refTop[0][i + 1] + e->u.ref.refLeft[0][i + 1]; dcVal = (hm_wave_sum_i(s) + 4) / 8; }
  HM_LVA(int32_t, rec, 16); HM_LVA(int32_t, lv, 16);
  HM_WAVE_FOR(k) {
    if (k < jobs) {
      const int mode = e->rdModeList[k >> 1], ts = k & 1;
      int scanType = SCAN_DIAG;                                      // getCoefScanIdx (4x4 intra luma)
      if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
      int cbf;
      const uint32_t dist = s4_eval(e, A, B, p, k, 0, org, ps, mode, ts, scanType, dcVal, HM_LVAK(rec, k), HM_LVAK(lv, k), &cbf);
      // bits: xGetIntraBitsQT :1038 = prediction mode (codeIntraDirLumaAng :636), cbf, coefficients
      uint32_t frac = commonFrac;
      int predIdx = -1;
      for (int i = 0; i < 3; i++) if (mode == e->mpmPreds[i]) predIdx = i;
      s4_bin(e, A, k, &frac, S4_MODE, predIdx != -1);
      frac += 32768u * (uint32_t)(predIdx == -1 ? 5 : (predIdx ? 2 : 1));
      s4_bin(e, A, k, &frac, S4_CBF, cbf);
      if (cbf) s4_code_coeff(e, A, B, k, k, scanType, ts, &frac);
      const uint32_t bits = frac >> 15;
      A->outDist[k] = dist; A->outBits[k] = bits; A->outCbf[k] = (uint8_t)cbf;
      A->outCost[k] = calc_rd_cost(e, bits, dist);
    }
  }
  HM_SYNC();
  // the reference's sequential decision: transform skip against the transform (xRecurIntraCodingQT :1452-1530), then candidate
  // against candidate (estIntraPredQT :2520-2560); strict "<" throughout
  double bestCost = HM_MAX_DOUBLE; int bestJob = 0;
  for (int c = 0; c < numModes; c++) {
    double single = A->outCost[2 * c]; int pick = 2 * c;
    if (A->outCbf[2 * c + 1]) { const double c1 = A->outCost[2 * c + 1]; if (c1 < single) { single = c1; pick = 2 * c + 1; } }
    if (single < bestCost) { bestCost = single; bestJob = pick; }
  }
  bestJob = HM_UNI(bestJob);
  const int bestMode = e->rdModeList[bestJob >> 1], bestCbf = A->outCbf[bestJob];
  e->outBits = (uint32_t)bestMode; e->outDistY = A->outDist[bestJob]; e->outRdCost = bestCost;
  HM_WAVE_FOR(k) {
    if (k == bestJob) {
      TCoeff *coef = e->cc + z * 16; Pel *ro = ws->reco + t->y * 64 + t->x;
#pragma unroll
      for (int i = 0; i < 16; i++) { coef[i] = HM_LVAK(lv, k)[i]; ro[(i >> 2) * 64 + (i & 3)] = (Pel)HM_LVAK(rec, k)[i]; }
    }
  }
  if (hm_lane() == 0) {
    m->tr[z] = (uint8_t)t->trDepth; m->cbf[0][z] = (uint8_t)(bestCbf << t->trDepth); m->ts[0][z] = (uint8_t)(bestJob & 1); m->dirL[z] = (uint8_t)bestMode;
  }
  HM_SYNC();
}

// ------------------------------------------------------------------------------------------------
// chroma of an 8x8 CU: its one 4x4 block per component under the five chroma modes (estIntraPredChromaQT :2698-2849 over
// xRecurIntraChromaCodingQT :1958-2145), ten evaluations at once: job 2 * modeIndex + (component - 1).  None of them codes a
// bin before the mode's bits are counted, so all start from the CU's entry snapshot in e->cur.  Then one lane per mode counts
// the mode's bits (chroma prediction mode, both cbfs, Cb coefficients, Cr coefficients: xGetIntraBitsQT :1038) on its own
// context copy, and the modes are compared in the reference's order.  `leaf`: the TU that carries the chroma blocks (the CU's
// root TU, or its first 4x4 luma quadrant when the luma transform is split).  Not used when that TU tries transform skip
// (chroma_tu: the Cr trial then starts from the contexts the Cb winner left, so the evaluations chain).
// Writes the winner (dirC / cbf / ts of the CU, coefficients, reconstruction into ws->reco) and returns its distortion.
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE uint32_t simt4_chroma_cu(Shared *e, TU leafv, int cuZ)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); leafv = hm_uni_struct(leafv);
  const TU *t = &leafv; CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const int zc = t->cuZ + t->cRelZ, r = hm_z2r(zc);
  const int x4 = e->ctuX * 16 + (r & 15), y4 = e->ctuY * 16 + (r >> 4), px = e->ctuX * 32 + t->cx, py = e->ctuY * 32 + t->cy;
  // reference lines: Cr into slot 1, Cb into slot 0 (unfiltered: chroma never uses the smoothed lines)
  init_adi_pattern(e, 2, px, py, 4, x4, y4, 2, 0);
  HM_PAR_FOR(i, 9) { e->u.ref.refTop[1][i] = e->u.ref.refTop[0][i]; e->u.ref.refLeft[1][i] = e->u.ref.refLeft[0][i]; }
  HM_SYNC();
  init_adi_pattern(e, 1, px, py, 4, x4, y4, 2, 0);
  int modeList[5] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX, DM_CHROMA_IDX};       // getAllowedChromaDir, TComDataCU.cpp:1486
  const int lumaDir = m->dirL[cuZ];
  for (int i = 0; i < 4; i++) if (lumaDir == modeList[i]) { modeList[i] = 34; break; }
  Simt4A *A = s4_a(e); Simt4B *B = s4_b(e);
  s4_setup(e, &e->cur, 1, 5 + t->trDepth, 5, C_CHROMA_PRED, 10);
  const S4Par p = s4_params(e, 1);
  const uint32_t baseFrac = (uint32_t)(e->cur.frac & 32767);
  int dcVal[2];
  for (int c = 0; c < 2; c++) { int s = 0; HM_PAR_FOR(i, 4) s += e->u.ref.refTop[c][i + 1] + e->u.ref.refLeft[c][i + 1]; dcVal[c] = (hm_wave_sum_i(s) + 4) / 8; }
  HM_LVA(int32_t, rec, 16); HM_LVA(int32_t, lv, 16);
  HM_WAVE_FOR(k) {
    if (k < 10) {
      const int c = k & 1, mi = k >> 1, ps = e->stride[1 + c];
      const int dirC = mi == 0 ? modeList[0] : (mi == 1 ? modeList[1] : (mi == 2 ? modeList[2] : (mi == 3 ? modeList[3] : modeList[4])));
      const int mode = dirC == DM_CHROMA_IDX ? lumaDir : dirC;
      int scanType = SCAN_DIAG;                                      // getCoefScanIdx (4x4 intra chroma)
      if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
      const Pel *org = e->fb.org[1 + c] + (size_t)py * ps + px;
      int cbf;
      const uint32_t sse = s4_eval(e, A, B, p, k, c, org, ps, mode, 0, scanType, c ? dcVal[1] : dcVal[0], HM_LVAK(rec, k), HM_LVAK(lv, k), &cbf);
      A->outDist[k] = (uint32_t)(e->fb.chromaWeight * (double)sse);   // getDistPart, TComRdCost.cpp:447-450
      A->outCbf[k] = (uint8_t)cbf;
    }
  }
  HM_SYNC();
  HM_WAVE_FOR(k) {
    if (k < 10 && !(k & 1)) {                                        // lane 2 * modeIndex: the bits of that mode
      const int mi = k >> 1;
      const int dirC = mi == 0 ? modeList[0] : (mi == 1 ? modeList[1] : (mi == 2 ? modeList[2] : (mi == 3 ? modeList[3] : modeList[4])));
      const int mode = dirC == DM_CHROMA_IDX ? lumaDir : dirC;
      int scanType = SCAN_DIAG;
      if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
      uint32_t frac = baseFrac;
      s4_bin(e, A, k, &frac, S4_MODE, dirC != DM_CHROMA_IDX);        // codeIntraDirChroma :692
      if (dirC != DM_CHROMA_IDX) frac += 2u * 32768u;
      const int cbfU = A->outCbf[k], cbfV = A->outCbf[k + 1];
      s4_bin(e, A, k, &frac, S4_CBF, cbfU);                          // xEncSubdivCbfQT :856: both cbfs at the CU's root TU
      s4_bin(e, A, k, &frac, S4_CBF, cbfV);
      if (cbfU) s4_code_coeff(e, A, B, k, k, scanType, 0, &frac);
      if (cbfV) s4_code_coeff(e, A, B, k, k + 1, scanType, 0, &frac);
      const uint32_t bits = frac >> 15, dist = A->outDist[k] + A->outDist[k + 1];
      A->outBits[k] = dist;
      A->outCost[k] = calc_rd_cost(e, bits, dist);
    }
  }
  HM_SYNC();
  double bestCost = HM_MAX_DOUBLE; int best = 0;
  for (int mi = 0; mi < 5; mi++) { const double c = A->outCost[2 * mi]; if (c < bestCost) { bestCost = c; best = mi; } }
  best = HM_UNI(best);
  const uint32_t bestDist = A->outBits[2 * best];
  const int cbfU = A->outCbf[2 * best], cbfV = A->outCbf[2 * best + 1];
  HM_WAVE_FOR(k) {
    if ((k >> 1) == best && k < 10) {
      const int po = HM_PLANE_OFF(1 + (k & 1));
      TCoeff *coef = e->cc + po + t->cOff; Pel *ro = ws->reco + po + t->cy * 32 + t->cx;
#pragma unroll
      for (int i = 0; i < 16; i++) { coef[i] = HM_LVAK(lv, k)[i]; ro[(i >> 2) * 32 + (i & 3)] = (Pel)HM_LVAK(rec, k)[i]; }
    }
  }
  { // decision arrays of the CU: cbf at the leaf's depth, merged into depth 0 when the luma transform is split (xRecurIntraChromaCodingQT :2120-2140)
    const int vU = t->trDepth ? 3 * cbfU : cbfU, vV = t->trDepth ? 3 * cbfV : cbfV, bm = best == 0 ? modeList[0] : (best == 1 ? modeList[1] : (best == 2 ? modeList[2] : (best == 3 ? modeList[3] : modeList[4])));
    HM_PAR_FOR(i, 4) { m->cbf[1][cuZ + i] = (uint8_t)vU; m->cbf[2][cuZ + i] = (uint8_t)vV; m->ts[1][cuZ + i] = 0; m->ts[2][cuZ + i] = 0; m->dirC[cuZ + i] = (uint8_t)bm; }
  }
  HM_SYNC();
  return bestDist;
}

// ------------------------------------------------------------------------------------------------
// One 4x4 luma transform block inside a residual quadtree (xIntraCodingTUBlock :1074 + xGetIntraBitsQT :1038 of a leaf of
// xRecurIntraCodingQT, no transform-skip trial: 2Nx2N CUs).  Nothing runs beside it -- the blocks of a quadtree chain through the
// reconstruction and the CABAC state -- but for a block this small the per-lane form of the evaluation (hm355_simt4.h) on ONE lane
// is shorter than the wave-uniform form with its 64-lane staging and its LDS / HBM round trips, so it runs there.  Side effects
// are those of the two reference functions: coefficients and reconstruction in the quadtree layer buffers and the picture,
// tr / cbf of the partition, the estimator (e->cur) advanced past the block's syntax.  Distortion / bits in e->outDistY / e->outBits.
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE void simt4_luma_leaf(Shared *e, TU tv)
{
  HM_ENTRY(e); tv = hm_uni_struct(tv);
  const TU *t = &tv; CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const int z = t->cuZ + t->relZ, ps = e->stride[0], r = hm_z2r(z), mode = m->dirL[z];
  init_adi_pattern(e, 0, e->ctuX * 64 + t->x, e->ctuY * 64 + t->y, 4, e->ctuX * 16 + (r & 15), e->ctuY * 16 + (r >> 4), 1, 0);
  { // the bins in front of the block's own (xEncIntraHeader :965), on the estimator itself
    CabacR cr; cabr_load(e, cr, &e->cur);
    cr.frac &= 32767;
    enc_intra_header(e, &cr, t, 1, 0);
    cabr_store(cr, &e->cur);
  }
  Simt4A *A = s4_a(e); Simt4B *B = s4_b(e);
  s4_setup(e, &e->cur, 0, 0, 0, C_INTRA_LUMA, 1);                  // luma cbf below the CU's root TU: context 0
  const S4Par p = s4_params(e, 0);
  const Pel *org = e->fb.org[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x;
  int dcVal;
  { int s = 0; HM_PAR_FOR(i, 4) s += e->u.ref.refTop[0][i + 1] + e->u.ref.refLeft[0][i + 1]; dcVal = (hm_wave_sum_i(s) + 4) / 8; }
  int scanType = SCAN_DIAG;
  if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
  const uint32_t frac0 = (uint32_t)e->cur.frac;
  HM_WAVE_FOR(k) {
    if (k == 0) {
      int32_t rec[16], lv[16]; int cbf;
      const uint32_t dist = s4_eval(e, A, B, p, 0, 0, org, ps, mode, 0, scanType, dcVal, rec, lv, &cbf);
      uint32_t frac = frac0;
      s4_bin(e, A, 0, &frac, S4_CBF, cbf);                          // xEncSubdivCbfQT :856: a 4x4 block codes no split flag
      if (cbf) s4_code_coeff(e, A, B, 0, 0, scanType, 0, &frac);
      A->outDist[0] = dist; A->outBits[0] = frac; A->outCbf[0] = (uint8_t)cbf;
      TCoeff *coef = ws->qtCoef[3] + z * 16; Pel *rq = ws->qtRec[3] + t->y * 64 + t->x;
      Pel *recPic = e->fb.rec[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x;
#pragma unroll
      for (int i = 0; i < 16; i++) { coef[i] = lv[i]; rq[(i >> 2) * 64 + (i & 3)] = (Pel)rec[i]; recPic[(i >> 2) * ps + (i & 3)] = (Pel)rec[i]; }
    }
  }
  HM_SYNC();
  // the estimator after the block: the contexts the lane advanced, the bit count
  HM_PAR_FOR(j, S4_MODE) {
    int c;
    if (j < S4_ONE) c = C_SIG + j; else if (j < S4_ABS) c = C_ONE + (j - S4_ONE); else if (j == S4_ABS) c = C_ABS;
    else if (j < S4_LY) c = C_LASTX + (j - S4_LX); else if (j < S4_CBF) c = C_LASTY + (j - S4_LY);
    else if (j == S4_CBF) c = C_QT_CBF; else c = C_TSKIP;
    e->cur.s[c] = A->ctx[j][0];
  }
  const int cbf = A->outCbf[0];
  e->cur.frac = (uint64_t)A->outBits[0];
  e->outDistY = A->outDist[0]; e->outBits = A->outBits[0] >> 15;
  if (hm_lane() == 0) { m->tr[z] = (uint8_t)t->trDepth; m->cbf[0][z] = (uint8_t)(cbf << t->trDepth); }
  HM_SYNC();
}

// The 4x4 chroma blocks (Cb, Cr) of one transform unit under the CU's current chroma mode, no transform-skip trial
// (xIntraCodingTUBlock :1074 twice, from xRecurIntraChromaCodingQT :1958): two independent evaluations from the same estimator
// state, one per lane.  Side effects as the reference function's; returns the sum of the weighted distortions.
HM_DEV HM_NOINLINE uint32_t simt4_chroma_leaf(Shared *e, TU tv)
{
  HM_ENTRY(e); tv = hm_uni_struct(tv);
  const TU *t = &tv; CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const int zc = t->cuZ + t->cRelZ, r = hm_z2r(zc), layer = 5 - t->log2;
  const int x4 = e->ctuX * 16 + (r & 15), y4 = e->ctuY * 16 + (r >> 4), px = e->ctuX * 32 + t->cx, py = e->ctuY * 32 + t->cy;
  init_adi_pattern(e, 2, px, py, 4, x4, y4, 2, 0);
  HM_PAR_FOR(i, 9) { e->u.ref.refTop[1][i] = e->u.ref.refTop[0][i]; e->u.ref.refLeft[1][i] = e->u.ref.refLeft[0][i]; }
  HM_SYNC();
  init_adi_pattern(e, 1, px, py, 4, x4, y4, 2, 0);
  int mode = m->dirC[zc];
  if (mode == DM_CHROMA_IDX) mode = m->dirL[zc & ~3];
  int scanType = SCAN_DIAG;
  if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
  Simt4A *A = s4_a(e); Simt4B *B = s4_b(e);
  s4_setup(e, &e->cur, 1, 5 + t->trDepth, 5, C_CHROMA_PRED, 2);
  const S4Par p = s4_params(e, 1);
  int dcVal[2];
  for (int c = 0; c < 2; c++) { int s = 0; HM_PAR_FOR(i, 4) s += e->u.ref.refTop[c][i + 1] + e->u.ref.refLeft[c][i + 1]; dcVal[c] = (hm_wave_sum_i(s) + 4) / 8; }
  HM_WAVE_FOR(k) {
    if (k < 2) {
      const int ps = e->stride[1 + k], po = HM_PLANE_OFF(1 + k);
      const Pel *org = e->fb.org[1 + k] + (size_t)py * ps + px;
      int32_t rec[16], lv[16]; int cbf;
      const uint32_t sse = s4_eval(e, A, B, p, k, k, org, ps, mode, 0, scanType, k ? dcVal[1] : dcVal[0], rec, lv, &cbf);
      A->outDist[k] = (uint32_t)(e->fb.chromaWeight * (double)sse);
      A->outCbf[k] = (uint8_t)cbf;
      TCoeff *coef = ws->qtCoef[layer] + po + t->cOff; Pel *rq = ws->qtRec[layer] + po + t->cy * 32 + t->cx;
      Pel *recPic = e->fb.rec[1 + k] + (size_t)py * ps + px;
#pragma unroll
      for (int i = 0; i < 16; i++) { coef[i] = lv[i]; rq[(i >> 2) * 32 + (i & 3)] = (Pel)rec[i]; recPic[(i >> 2) * ps + (i & 3)] = (Pel)rec[i]; }
    }
  }
  HM_SYNC();
  const int cbfU = A->outCbf[0], cbfV = A->outCbf[1];
  const uint32_t dist = A->outDist[0] + A->outDist[1];
  HM_PAR_FOR(i, t->cParts) { m->cbf[1][zc + i] = (uint8_t)(cbfU << t->trDepth); m->cbf[2][zc + i] = (uint8_t)(cbfV << t->trDepth); m->ts[1][zc + i] = 0; m->ts[2][zc + i] = 0; }
  HM_SYNC();
  return dist;
}
