// hm355 -- candidates in lanes, 8x8 blocks: the first pass of estIntraPredQT (:2473-2490) over the RD candidates of an 8x8
// luma prediction unit.  Every candidate is the same evaluation of one 8x8 transform block from the same CABAC snapshot
// (xRecurIntraCodingQT with bCheckFirst: no split), and only its cost is kept -- the winner is evaluated again by the closing
// pass with the full residual quadtree.  So the wavefront alternates between two shapes of parallelism:
//   * sample work (prediction, residual, 8-point transforms, de-quantisation, reconstruction, SSE) one candidate after the
//     other with the 64 lanes on the block's 64 samples,
//   * the serial chains (RDOQ level decision, CABAC bit estimate) for all candidates at once, one candidate per lane.
// Arithmetic and operation order are the reference's throughout (see rdoq / code_coeff_nxn in hm355_core.h, of which the
// per-lane functions here are the one-candidate-per-lane form).  Included from hm355_core.h after hm355_simt4.h.
#pragma once

#define HM_S8 11                          // candidates per batch: 8 by SATD + up to 3 most probable modes
enum { S8_SIG = 0, S8_ONE = 21, S8_ABS = 37, S8_LX = 41, S8_LY = 44, S8_CG = 47, S8_CBF = 49, S8_MODE = 50, S8_NCTX = 52 };
enum { T8_SIG = 0, T8_ONE = 42, T8_ABS = 74, T8_LASTX = 82, T8_LASTY = 88, T8_CG = 94, T8_CBF = 98, T8_N = 100 };
struct Simt8A {                           // overlays Shared::bufA
  int32_t tile[2][64];                    // transform stages of the candidate under the lanes
  int32_t tab[T8_N];                      // bit costs of the start state
  double outCost[HM_S8];
  uint32_t outDist[HM_S8], outFrac[HM_S8];
  int16_t cs[64][HM_S8];                  // coefficients in scan order
  uint8_t ctx[S8_NCTX][HM_S8];            // context states of each candidate
  uint8_t outCbf[HM_S8], pad[5];
  uint8_t lps[128];
};
struct Simt8B {                           // overlays Shared::u behind the reference sample lines
  int32_t dc[64][HM_S8];                  // level at decision time (low half) | final signed level (high half), scan order
  uint8_t scan[3][64];                    // scan position -> raster position
  uint8_t scanCG[3][4];
};
static_assert(sizeof(Simt8A) <= sizeof(((Shared *)0)->bufA), "Simt8A overlays bufA");
static_assert(offsetof(RefLds, refMain) + sizeof(Simt8B) <= sizeof(((Shared *)0)->u), "Simt8B overlays the tail of the LDS union");
HM_DEV inline Simt8A *s8_a(Shared *e) { return (Simt8A *)e->bufA; }
HM_DEV inline Simt8B *s8_b(Shared *e) { return (Simt8B *)((char *)&e->u + offsetof(RefLds, refMain)); }

struct S8Par { int chroma, bitDepth, qBits, quantCoef; double errScale, lambda; int64_t rdFactor; int dqShift, dqScale, dqMin, dqMax; };
HM_DEV inline S8Par s8_params(const Shared *e, int chroma)
{
  S8Par p;
  const int tshift = 15 - e->bitDepth - 3;
  p.chroma = chroma;
  p.bitDepth = e->bitDepth; p.qBits = 14 + e->fb.qpPer[chroma] + tshift; p.quantCoef = HM_QUANT_SCALES[e->fb.qpRem[chroma]];
  p.errScale = e->fb.errScale[chroma][1]; p.lambda = chroma ? e->fb.lambdaC : e->fb.lambda; p.rdFactor = e->fb.rdFactor[chroma];
  p.dqShift = 6 - (tshift + e->fb.qpPer[chroma]); p.dqScale = HM_INV_QUANT_SCALES[e->fb.qpRem[chroma]];
  int tgt = 25 + p.dqShift; if (tgt > 16) tgt = 16;
  p.dqMin = -(1 << (tgt - 1)); p.dqMax = (1 << (tgt - 1)) - 1;
  return p;
}

// bit costs of the start state (estBit, TEncSbac.cpp:1717-1956), per-candidate context copies, scans of an 8x8 block of one component type.
// Context numbering inside the tables / copies: significance contexts by their increment (0..20), greater-than-1 contexts 4 * set + c1 with
// set = 0..3 for luma and 0..1 for chroma (whose sets are 4, 5 in the reference's numbering), greater-than-2 by set, the three last-position
// contexts an 8x8 block uses per coordinate, the two coded-sub-block contexts.  cbfCtx / cbfCodeCtx: the cbf context RDOQ prices with / the one
// the syntax codes with (indices inside C_QT_CBF); modeCtx: context of the prediction-mode bin.
HM_DEV inline void s8_setup(Shared *e, const Cabac *cb, int jobs, int chroma, int cbfCtx, int cbfCodeCtx, int modeCtx)
{
  Simt8A *A = s8_a(e); Simt8B *B = s8_b(e);
  const int sigOff = C_SIG + (chroma ? 28 : 0), oneOff = C_ONE + (chroma ? 16 : 0), absOff = C_ABS + (chroma ? 4 : 0);
  const int lxOff = chroma ? C_LASTX + 15 : C_LASTX + 3, lyOff = chroma ? C_LASTY + 15 : C_LASTY + 3, cgOff = C_SIG_CG + (chroma ? 2 : 0);
  HM_PAR_FOR(i, T8_N) {
    int v = 0;
    if (i < T8_ONE) v = (chroma && (i >> 1) >= 16) ? 0 : HM_LT()->ebits[cb->s[sigOff + (i >> 1)] ^ (i & 1)];
    else if (i < T8_ABS) v = (chroma && ((i - T8_ONE) >> 1) >= 8) ? 0 : HM_LT()->ebits[cb->s[oneOff + ((i - T8_ONE) >> 1)] ^ (i & 1)];
    else if (i < T8_LASTX) v = (chroma && ((i - T8_ABS) >> 1) >= 2) ? 0 : HM_LT()->ebits[cb->s[absOff + ((i - T8_ABS) >> 1)] ^ (i & 1)];
    else if (i < T8_CG) { // last-position group index g = 0..5 of an 8x8 block: context (g >> 1) of the block's three; g > 3 adds one bypass bit (xGetRateLast :2815)
      const int g = (i - T8_LASTX) % 6, off = i < T8_LASTY ? lxOff : lyOff;
      for (int c = 0; c < g; c++) v += HM_LT()->ebits[cb->s[off + (c >> 1)] ^ 1];
      if (g < 5) v += HM_LT()->ebits[cb->s[off + (g >> 1)] ^ 0];
      if (g > 3) v += 32768 * ((g - 2) >> 1);
    } else if (i < T8_CBF) v = HM_LT()->ebits[cb->s[cgOff + ((i - T8_CG) >> 1)] ^ (i & 1)];
    else v = HM_LT()->ebits[cb->s[C_QT_CBF + cbfCtx] ^ (i & 1)];
    A->tab[i] = v;
  }
  HM_PAR_FOR(i, 128) A->lps[i] = HM_NEXT_LPS[i];
  HM_PAR_FOR(i, S8_NCTX * HM_S8) {
    const int j = i / HM_S8, k = i - j * HM_S8;
    int c;
    if (j < S8_ONE) c = sigOff + (chroma && j >= 16 ? 0 : j); else if (j < S8_ABS) c = oneOff + (chroma ? ((j - S8_ONE) & 7) : (j - S8_ONE)); else if (j < S8_LX) c = absOff + (chroma ? ((j - S8_ABS) & 1) : (j - S8_ABS));
    else if (j < S8_LY) c = lxOff + (j - S8_LX); else if (j < S8_CG) c = lyOff + (j - S8_LY);
    else if (j < S8_CBF) c = cgOff + (j - S8_CG); else if (j == S8_CBF) c = C_QT_CBF + cbfCodeCtx; else c = modeCtx;
    if (k < jobs) A->ctx[j][k] = cb->s[c];
  }
  HM_PAR_FOR(i, 192) { const int ty = i >> 6, sp = i & 63; B->scan[ty][sp] = (uint8_t)e->tab->scan[ty][1][sp]; }
  HM_PAR_FOR(i, 12) { const int ty = i >> 2, cg = i & 3; B->scanCG[ty][cg] = (uint8_t)e->tab->scanCG[ty][1][cg]; }
  HM_SYNC();
}
HM_DEV inline void s8_bin(const Shared *e, Simt8A *A, int k, uint32_t *frac, int c, int bin)
{
  const int st = A->ctx[c][k];
  *frac += (uint32_t)HM_LT()->ebits[st ^ bin];
  A->ctx[c][k] = (uint8_t)(bin == (st & 1) ? (st < 124 ? st + 2 : st) : A->lps[st]);
}
// xGetICRate, TComTrQuant.cpp:2725-2800
HM_FINL int s8_ic_rate(const int32_t *tab, uint32_t absLevel, int ctxOne, int ctxAbs, int goRice, int c1Idx, int c2Idx)
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
    if (c1Idx < 8) { rate += tab[T8_ONE + ctxOne * 2 + 1]; if (c2Idx < 1) rate += tab[T8_ABS + ctxAbs * 2 + 1]; }
  } else if (absLevel == 1) rate += tab[T8_ONE + ctxOne * 2];
  else if (absLevel == 2) { rate += tab[T8_ONE + ctxOne * 2 + 1]; rate += tab[T8_ABS + ctxAbs * 2]; }
  else rate = 0;
  return rate;
}
HM_FINL int32_t s8_level_double(int sc, const S8Par &p)
{
  const int64_t cap = 2147483647LL - (1LL << (p.qBits - 1));
  const int64_t tl = (int64_t)hm_abs(sc) * p.quantCoef;
  return (int32_t)(tl < cap ? tl : cap);
}
// significance context increment of scan position sp of candidate k's block, given the flags of the coefficient groups decided so far
HM_FINL int s8_sig_idx(const Simt8B *B, int scanType, int firstCtx, int sp, int cgMask, int chroma)
{
  const int cgBlkPos = B->scanCG[scanType][sp >> 4], cgx = cgBlkPos & 1, cgy = cgBlkPos >> 1;
  const int sigRight = cgx < 1 ? ((cgMask >> (cgBlkPos + 1)) & 1) : 0, sigLower = cgy < 1 ? ((cgMask >> (cgBlkPos + 2)) & 1) : 0;
  return sig_ctx_inc(sigRight + (sigLower << 1), firstCtx, B->scan[scanType][sp], 3, chroma);
}

// RDOQ of candidate k's 8x8 block (TComTrQuant::xRateDistOptQuant, TComTrQuant.cpp:1974-2511): coefficients in A->cs[.][k]
// (scan order); leaves the signed levels in the high halves of B->dc[.][k]; returns the sum of their magnitudes.
HM_DEV inline int s8_rdoq(Simt8A *A, Simt8B *B, const S8Par &p, int k, int scanType)
{
  const int32_t *tab = A->tab;
  const int qBits = p.qBits, half = 1 << (qBits - 1);
  const double lambda = p.lambda, errScale = p.errScale;
  const int firstCtx = first_sig_ctx(8, scanType, p.chroma), lumaSets = p.chroma ? 0 : 2;      // luma: coefficient groups behind the first use context sets 2, 3
  double blockUncoded = 0, baseCost = 0;
  double cgSig0 = 0, cgSig1 = 0, cgSig2 = 0, cgSig3 = 0;          // cost of the coded-sub-block flag per group (scan order)
  int last = -1, cgLast = -1, ctxSet = 0, c1 = 1, c2 = 0, c1Idx = 0, c2Idx = 0, goRice = 0;
  int cgMask = 0, cgSets = 0;                                     // group flags (bit = raster position of the group); context set each group started with
  for (int cg = 3; cg >= 0; cg--) {
    const int cgBlkPos = B->scanCG[scanType][cg], cgx = cgBlkPos & 1, cgy = cgBlkPos >> 1, cgBit = 1 << cgBlkPos;
    const int sigRight = cgx < 1 ? ((cgMask >> (cgBlkPos + 1)) & 1) : 0, sigLower = cgy < 1 ? ((cgMask >> (cgBlkPos + 2)) & 1) : 0;
    const int pattern = sigRight + (sigLower << 1);
    double sigCost = 0, sigCost0 = 0, codedLevelAndDist = 0, uncodedDist = 0; int nnzBeforePos0 = 0;
    cgSets |= ctxSet << (2 * cg);
    for (int q = 15; q >= 0; q--) {
      const int sp = cg * 16 + q;
      const int32_t lvlD = s8_level_double(A->cs[sp][k], p);
      uint32_t mx = (uint32_t)((lvlD + half) >> qBits); if (mx > 32767u) mx = 32767u;
      const double err = (double)lvlD, c0 = err * err * errScale;
      blockUncoded += c0;
      if (mx > 0 && last < 0) { last = sp; cgLast = cg; ctxSet = cg > 0 ? lumaSets : 0; cgSets = ctxSet << (2 * cg); }
      uint32_t level = 0; double cSig = 0, cCoeff = c0;
      if (last >= 0) {
        const int isLast = (sp == last);
        const int si = isLast ? 0 : sig_ctx_inc(pattern, firstCtx, B->scan[scanType][sp], 3, p.chroma);
        if (!isLast && mx < 3) { cSig = lambda * (double)tab[T8_SIG + si * 2]; cCoeff = c0 + cSig; }      // xGetCodedLevel :2660
        else cCoeff = HM_MAX_DOUBLE;
        if (mx > 0) {
          double currCostSig = 0;
          if (!isLast) currCostSig = lambda * (double)tab[T8_SIG + si * 2 + 1];
          const uint32_t minAbs = mx > 1 ? mx - 1 : 1;
          for (int al = (int)mx; al >= (int)minAbs; al--) {
            const double de = (double)(lvlD - (int32_t)((uint32_t)al << qBits));
            const double dist = de * de * errScale;
            const double rc = lambda * (double)s8_ic_rate(tab, (uint32_t)al, 4 * ctxSet + c1, ctxSet, goRice, c1Idx, c2Idx);
            double cc = dist + rc;
            cc += currCostSig;
            if (cc < cCoeff) { level = (uint32_t)al; cCoeff = cc; cSig = currCostSig; }
          }
        }
        baseCost += cCoeff;
        const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel && level > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
        if (q == 0 && cg > 0) { ctxSet = ((cg - 1) > 0 ? lumaSets : 0) + (c1 == 0 ? 1 : 0); c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0; }
      } else baseCost += c0;
      sigCost += cSig;
      if (q == 0) sigCost0 = cSig;
      if (level) {
        cgMask |= cgBit;
        codedLevelAndDist += cCoeff - cSig;
        uncodedDist += c0;
        if (q != 0) nnzBeforePos0++;
      }
      B->dc[sp][k] = (int32_t)level;
    }
    if (cgLast >= 0) {
      if (cg) {
        const int cgCtx = ((sigRight + sigLower) != 0) * 2;                              // getSigCoeffGroupCtxInc :2872
        double cgs = 0;
        if (!(cgMask & cgBit)) {
          const double r0 = lambda * (double)tab[T8_CG + cgCtx];
          baseCost += r0 - sigCost;
          cgs = r0;
        } else if (cg < cgLast) {
          if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
          double costZeroCG = baseCost;
          const double r0 = lambda * (double)tab[T8_CG + cgCtx], r1 = lambda * (double)tab[T8_CG + cgCtx + 1];
          baseCost += r1;
          costZeroCG += r0;
          cgs = r1;
          costZeroCG += uncodedDist; costZeroCG -= codedLevelAndDist; costZeroCG -= sigCost;
          if (costZeroCG < baseCost) {
            cgMask &= ~cgBit; baseCost = costZeroCG; cgs = r0;
            for (int q = 15; q >= 0; q--) B->dc[cg * 16 + q][k] |= 0x40000000;      // zeroed: the decision-time level stays for the walks below
          }
        }
        if (cg == 1) cgSig1 = cgs; else if (cg == 2) cgSig2 = cgs; else cgSig3 = cgs;
      } else cgMask |= cgBit;
    }
  }
  (void)cgSig0;
  if (last < 0) return 0;
  double bestCost = blockUncoded + lambda * (double)tab[T8_CBF];            // TComTrQuant.cpp:2310-2316
  baseCost += lambda * (double)tab[T8_CBF + 1];
  int bestLastP1 = 0, found = 0;
  for (int cg = cgLast; cg >= 0 && !found; cg--) {
    const int cgBlkPos = B->scanCG[scanType][cg];
    baseCost -= (cg == 0 ? 0.0 : (cg == 1 ? cgSig1 : (cg == 2 ? cgSig2 : cgSig3)));
    if (!((cgMask >> cgBlkPos) & 1)) continue;
    // the decision chain's state inside this group, walked again to price the levels it decided
    const int wSet = (cgSets >> (2 * cg)) & 3; int wC1 = 1, wC1Idx = 0, wC2Idx = 0, wGoR = 0;
    for (int q = (cg == cgLast ? (last & 15) : 15); q >= 0; q--) {
      const int sp = cg * 16 + q, lev = B->dc[sp][k] & 0xffff;
      const int si = sp == last ? 0 : s8_sig_idx(B, scanType, firstCtx, sp, cgMask, p.chroma);
      if (lev) {
        const int blkPos = B->scan[scanType][sp];
        int py = blkPos >> 3, px = blkPos & 7;
        if (scanType == SCAN_VER) { const int t = px; px = py; py = t; }
        const double costLast = lambda * (double)(tab[T8_LASTX + hm_group_idx(px)] + tab[T8_LASTY + hm_group_idx(py)]);
        const double cSig = (sp == last) ? 0.0 : lambda * (double)tab[T8_SIG + si * 2 + 1];
        const double t1 = baseCost + costLast;
        const double totalCost = t1 - cSig;
        if (totalCost < bestCost) { bestLastP1 = sp + 1; bestCost = totalCost; }
        if (lev > 1) { found = 1; break; }
        const int32_t lvlD = s8_level_double(A->cs[sp][k], p);
        const double err = (double)lvlD;
        const double de = (double)(lvlD - (int32_t)((uint32_t)lev << qBits));
        const double dist = de * de * errScale;
        const double rc = lambda * (double)s8_ic_rate(tab, (uint32_t)lev, 4 * wSet + wC1, wSet, wGoR, wC1Idx, wC2Idx);
        double cc = dist + rc;
        cc += cSig;
        baseCost -= cc; baseCost += err * err * errScale;
        { // the chain's update for this level (a 1: levels above 1 left the loop)
          const uint32_t baseLevel = (wC1Idx < 8) ? (2 + (wC2Idx < 1)) : 1;
          if ((uint32_t)lev >= baseLevel && (uint32_t)lev > (3u << wGoR)) wGoR = wGoR + 1 < 4 ? wGoR + 1 : 4;
          wC1Idx++;
          if (wC1 < 3 && wC1 > 0) wC1++;
        }
      } else baseCost -= lambda * (double)tab[T8_SIG + si * 2];
    }
  }
  // levels with signs: zeroed groups and everything behind the chosen last position become 0
  int absSum = 0;
  for (int sp = 0; sp < 64; sp++) {
    const int v = B->dc[sp][k], dec = v & 0xffff, lv = (sp < bestLastP1 && !(v & 0x40000000)) ? dec : 0;
    absSum += lv;
    B->dc[sp][k] = dec | (int32_t)((uint32_t)(A->cs[sp][k] < 0 ? -lv : lv) << 16);
  }
  // sign bit hiding, TComTrQuant.cpp:2380-2510
  if (absSum >= 2) {
    int lastCG = -1;
    for (int subSet = 3; subSet >= 0; subSet--) {
      const int subPos = subSet << 4;
      int lastNZ = -1, firstNZ = 16, parity = 0;
      for (int q = 0; q < 16; q++) { const int lv = B->dc[subPos + q][k] >> 16; if (lv) { lastNZ = q; if (firstNZ == 16) firstNZ = q; } parity ^= lv & 1; }
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4) {
        const uint32_t signbit = (B->dc[subPos + firstNZ][k] >> 16) > 0 ? 0 : 1;
        if (signbit != (uint32_t)parity) {
          const int64_t I64MAX = 0x7fffffffffffffffLL;
          int64_t minCostInc = I64MAX, curCost = I64MAX; int minK = -1, finalChange = 0, curChange = 0;
          const int wSet = (cgSets >> (2 * subSet)) & 3; int wC1 = 1, wC1Idx = 0, wC2Idx = 0, wGoR = 0;
          const int top = (subPos + 15 <= last) ? 15 : (last - subPos), kStart = (lastCG == 1 ? lastNZ : 15);
          for (int q = top; q >= 0; --q) {
            const int v = B->dc[subPos + q][k]; const uint32_t dec = (uint32_t)(v & 0xffff); const int dv = v >> 16;
            const int ctxOne = 4 * wSet + wC1, goR = wGoR, c1I = wC1Idx, c2I = wC2Idx;
            {
              const uint32_t baseLevel = (wC1Idx < 8) ? (2 + (wC2Idx < 1)) : 1;
              if (dec >= baseLevel && dec > (3u << wGoR)) wGoR = wGoR + 1 < 4 ? wGoR + 1 : 4;
              if (dec >= 1) wC1Idx++;
              if (dec > 1) { wC1 = 0; wC2Idx++; }
              else if (wC1 < 3 && wC1 > 0 && dec) wC1++;
            }
            if (q > kStart) continue;
            const int sc = A->cs[subPos + q][k];
            const int32_t lvlD = s8_level_double(sc, p);
            const int32_t deltaU = (int32_t)((lvlD - (int32_t)(dec << qBits)) >> (qBits - 8));
            const int si = (subPos + q == last) ? 0 : s8_sig_idx(B, scanType, firstCtx, subPos + q, cgMask, p.chroma);
            const int sigRateDelta = (subPos + q == last) ? 0 : tab[T8_SIG + si * 2 + 1] - tab[T8_SIG + si * 2];
            int rateIncUp, rateIncDown = 0;
            if (dec > 0) {
              const int rateNow = s8_ic_rate(tab, dec, ctxOne, wSet, goR, c1I, c2I);
              rateIncUp = s8_ic_rate(tab, dec + 1, ctxOne, wSet, goR, c1I, c2I) - rateNow;
              rateIncDown = s8_ic_rate(tab, dec - 1, ctxOne, wSet, goR, c1I, c2I) - rateNow;
            } else rateIncUp = tab[T8_ONE + ctxOne * 2];
            if (dv != 0) {
              const int64_t costUp = p.rdFactor * (-deltaU) + rateIncUp;
              int64_t costDown = p.rdFactor * (deltaU) + rateIncDown - ((hm_abs(dv) == 1) ? sigRateDelta : 0);
              if (lastCG == 1 && lastNZ == q && hm_abs(dv) == 1) costDown -= (4 << 15);
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
            const int v = B->dc[subPos + minK][k]; int mv = v >> 16;
            if (mv == 32767 || mv == -32768) finalChange = -1;
            mv = (A->cs[subPos + minK][k] < 0) ? mv - finalChange : mv + finalChange;
            B->dc[subPos + minK][k] = (v & 0xffff) | (int32_t)((uint32_t)mv << 16);
          }
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return absSum;
}

// TEncSbac::codeCoeffNxN, TEncSbac.cpp:1172-1525, of an 8x8 block on candidate k's private contexts (bits only); the levels are those of
// candidate `src` (the lane that decided them)
HM_DEV inline void s8_code_coeff(const Shared *e, Simt8A *A, const Simt8B *B, int k, int src, int chroma, int scanType, uint32_t *frac)
{
  int last = -1, cgMask = 0;
  for (int sp = 0; sp < 64; sp++)
    if ((B->dc[sp][src] >> 16) != 0) { last = sp; cgMask |= 1 << B->scanCG[scanType][sp >> 4]; }
  { // codeLastSignificantXY :1106
    const int blkPos = B->scan[scanType][last];
    int py = blkPos >> 3, px = blkPos & 7;
    if (scanType == SCAN_VER) { const int t = px; px = py; py = t; }
    const int gx = hm_group_idx(px), gy = hm_group_idx(py);
    int q;
    for (q = 0; q < gx; q++) s8_bin(e, A, k, frac, S8_LX + (q >> 1), 1);
    if (gx < 5) s8_bin(e, A, k, frac, S8_LX + (q >> 1), 0);
    for (q = 0; q < gy; q++) s8_bin(e, A, k, frac, S8_LY + (q >> 1), 1);
    if (gy < 5) s8_bin(e, A, k, frac, S8_LY + (q >> 1), 0);
    if (gx > 3) *frac += 32768u * (uint32_t)((gx - 2) >> 1);
    if (gy > 3) *frac += 32768u * (uint32_t)((gy - 2) >> 1);
  }
  const int firstCtx = first_sig_ctx(8, scanType, chroma), lastSet = last >> 4;
  int c1 = 1;
  for (int subSet = lastSet; subSet >= 0; subSet--) {
    const int subPos = subSet << 4, isLastSet = subSet == lastSet;
    const int cgBlkPos = B->scanCG[scanType][subSet], cgx = cgBlkPos & 1, cgy = cgBlkPos >> 1;
    const int sigRight = cgx < 1 ? ((cgMask >> (cgBlkPos + 1)) & 1) : 0, sigLower = cgy < 1 ? ((cgMask >> (cgBlkPos + 2)) & 1) : 0;
    if (isLastSet || subSet == 0) cgMask |= 1 << cgBlkPos;
    else s8_bin(e, A, k, frac, S8_CG + ((sigRight + sigLower) != 0), (cgMask >> cgBlkPos) & 1);
    if (!((cgMask >> cgBlkPos) & 1)) continue;
    const int pattern = sigRight + (sigLower << 1);
    const int top = isLastSet ? (last & 15) : 15;
    int numNonZero = isLastSet ? 1 : 0, firstNZ = isLastSet ? top : 16, lastNZ = isLastSet ? top : -1;
    for (int q = isLastSet ? top - 1 : 15; q >= 0; q--) {           // significance flags; the last coefficient itself is implied
      const int sig = (B->dc[subPos + q][src] >> 16) != 0;
      if (q > 0 || subSet == 0 || numNonZero) s8_bin(e, A, k, frac, S8_SIG + sig_ctx_inc(pattern, firstCtx, B->scan[scanType][subPos + q], 3, chroma), sig);
      if (sig) { numNonZero++; firstNZ = q; if (lastNZ < 0) lastNZ = q; }
    }
    if (numNonZero > 0) {
      const int signHidden = (lastNZ - firstNZ >= 4);
      const int ctxSet = ((subSet > 0 && !chroma) ? 2 : 0) + (c1 == 0 ? 1 : 0);
      c1 = 1;
      int firstC2 = -1, escape = 0, idx = 0;
      for (int q = lastNZ; q >= 0 && idx < 8; q--) {
        const int a = hm_abs(B->dc[subPos + q][src] >> 16);
        if (!a) continue;
        const int sym = a > 1;
        s8_bin(e, A, k, frac, S8_ONE + 4 * ctxSet + c1, sym);
        if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = q; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
        idx++;
      }
      if (c1 == 0 && firstC2 != -1) { const int sym = hm_abs(B->dc[subPos + firstC2][src] >> 16) > 2; s8_bin(e, A, k, frac, S8_ABS + ctxSet, sym); if (sym) escape = 1; }
      escape = escape || (numNonZero > 8);
      *frac += 32768u * (uint32_t)(signHidden ? numNonZero - 1 : numNonZero);
      if (escape) {
        int firstCoeff2 = 1; uint32_t goRice = 0; idx = 0;
        for (int q = lastNZ; q >= 0; q--) {
          const int a = hm_abs(B->dc[subPos + q][src] >> 16);
          if (!a) continue;
          const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (a >= baseLevel) {                                     // xWriteCoefRemainExGolomb :337
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
  }
}

// ------------------------------------------------------------------------------------------------
// The first pass over the RD candidates e->rdModeList[0..numModes) of the 8x8 PU `tv` (a 2Nx2N CU of depth 3).  e->cur holds the
// CU's entry snapshot, e->u.ref the PU's reference samples (both the plain and the smoothed lines), e->mpmPreds its most probable
// modes.  Returns the winning mode: the candidate of least cost, the first one on a tie (estIntraPredQT :2520-2560).
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE int simt8_luma_first_pass(Shared *e, TU tv, int numModes)
{
  HM_ENTRY(e); numModes = HM_UNI(numModes); tv = hm_uni_struct(tv);
  const TU *t = &tv;
  const int ps = e->stride[0], bitDepth = e->bitDepth;
  Simt8A *A = s8_a(e); Simt8B *B = s8_b(e);
  uint32_t commonFrac;                                             // bins every candidate codes alike (xEncIntraHeader :965, xEncSubdivCbfQT :856)
  {
    CabacR r; cabr_load(e, r, &e->cur);
    r.frac &= 32767;
    if (e->im) { code_skip_flag(e, &r, t->cuZ); enc_bin(e, &r, C_PRED_MODE, 1); }
    enc_bin(e, &r, C_PART, 1);                                     // 2Nx2N at the smallest CU size
    enc_bin(e, &r, C_SUBDIV + 2, 0);                               // transform_split of the 8x8 root TU: not split in this pass
    commonFrac = (uint32_t)r.frac;
  }
  s8_setup(e, &e->cur, numModes, 0, 1, 1, C_INTRA_LUMA);            // luma cbf at the CU's root TU: context 1
  const S8Par p = s8_params(e, 0);
  const Pel *org = e->fb.org[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x;
  int dcVal;
  { int s = 0; HM_PAR_FOR(i, 8) s += e->u.ref.refTop[0][i + 1] + e->u.ref.refLeft[0][i + 1]; dcVal = (hm_wave_sum_i(s) + 8) / 16; }
  const int s1 = 3 + bitDepth + 6 - 15, a1 = 1 << (s1 - 1), s2 = 9, a2 = 256;
  // ---- residual + forward transform of every candidate, lanes on the samples; coefficients to A->cs in the candidate's scan order
  for (int c = 0; c < numModes; c++) {
    const int mode = e->rdModeList[c];
    int scanType = SCAN_DIAG;
    if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
    HM_PAR_FOR(l, 64) { const int y = l >> 3, x = l & 7; A->tile[0][l] = org[y * ps + x] - pred_sample(e, mode, 8, 3, x, y, dcVal, bitDepth); }
    HM_SYNC();
    HM_PAR_FOR(l, 64) { // xTrMxN :836, first stage
      const int j = l >> 3, kk = l & 7; int32_t acc = 0;
      for (int i = 0; i < 8; i++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[0][j * 8 + i];
      A->tile[1][kk * 8 + j] = (acc + a1) >> s1;
    }
    HM_SYNC();
    HM_PAR_FOR(l, 64) { // second stage, one lane per scan position
      const int blkPos = B->scan[scanType][l], kk = blkPos >> 3, j = blkPos & 7; int32_t acc = 0;
      for (int i = 0; i < 8; i++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[1][j * 8 + i];
      A->cs[l][c] = (int16_t)((acc + a2) >> s2);
    }
    HM_SYNC();
  }
  // ---- level decision of every candidate, one per lane
  HM_WAVE_FOR(k) {
    if (k < numModes) {
      const int mode = e->rdModeList[k];
      int scanType = SCAN_DIAG;
      if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
      A->outCbf[k] = (uint8_t)(s8_rdoq(A, B, p, k, scanType) > 0);
    }
  }
  HM_SYNC();
  // ---- reconstruction + distortion of every candidate, lanes on the samples
  const int maxv = (1 << bitDepth) - 1, shiftSse = (bitDepth - 8) << 1, is2 = 20 - bitDepth;
  for (int c = 0; c < numModes; c++) {
    const int mode = e->rdModeList[c], cbf = A->outCbf[c];
    int scanType = SCAN_DIAG;
    if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
    if (cbf) {
      HM_PAR_FOR(l, 64) { // xDeQuant (flat) :1276 into raster order
        const int lvv = hm_clip3(p.dqMin, p.dqMax, B->dc[l][c] >> 16);
        int v;
        if (p.dqShift > 0) v = (lvv * p.dqScale + (1 << (p.dqShift - 1))) >> p.dqShift;
        else v = (int)((unsigned)(lvv * p.dqScale) << (-p.dqShift));
        A->tile[0][B->scan[scanType][l]] = hm_clip3(-32768, 32767, v);
      }
      HM_SYNC();
      HM_PAR_FOR(l, 64) { // xITrMxN :894
        const int j = l >> 3, i = l & 7; int32_t acc = 0;
        for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[0][kk * 8 + j];
        A->tile[1][j * 8 + i] = hm_clip3(-32768, 32767, (acc + 64) >> 7);
      }
      HM_SYNC();
    }
    uint32_t sse = 0;
    HM_PAR_FOR(l, 64) {
      const int j = l >> 3, i = l & 7; int resi = 0;
      if (cbf) {
        int32_t acc = 0;
        for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[1][kk * 8 + j];
        resi = hm_clip3(-32768, 32767, (acc + (1 << (is2 - 1))) >> is2);
      }
      const int r = hm_clip3(0, maxv, pred_sample(e, mode, 8, 3, i, j, dcVal, bitDepth) + resi);
      const int d = org[j * ps + i] - r; sse += (uint32_t)((d * d) >> shiftSse);
    }
    const uint32_t dist = hm_wave_sum(sse);
    if (hm_lane() == 0) A->outDist[c] = dist;
    HM_SYNC();
  }
  // ---- bits and cost of every candidate, one per lane (xGetIntraBitsQT :1038)
  HM_WAVE_FOR(k) {
    if (k < numModes) {
      const int mode = e->rdModeList[k], cbf = A->outCbf[k];
      int scanType = SCAN_DIAG;
      if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
      uint32_t frac = commonFrac;
      int predIdx = -1;
      for (int i = 0; i < 3; i++) if (mode == e->mpmPreds[i]) predIdx = i;
      s8_bin(e, A, k, &frac, S8_MODE, predIdx != -1);
      frac += 32768u * (uint32_t)(predIdx == -1 ? 5 : (predIdx ? 2 : 1));
      s8_bin(e, A, k, &frac, S8_CBF, cbf);
      if (cbf) s8_code_coeff(e, A, B, k, k, 0, scanType, &frac);
      A->outFrac[k] = frac;
      A->outCost[k] = calc_rd_cost(e, frac >> 15, A->outDist[k]);
    }
  }
  HM_SYNC();
  double bestCost = HM_MAX_DOUBLE; int best = 0;
  for (int c = 0; c < numModes; c++) { const double v = A->outCost[c]; if (v < bestCost) { bestCost = v; best = c; } }
  best = HM_UNI(best);
  const int bestMode = e->rdModeList[best];
  e->s8Winner = best;                                              // simt8_luma_winner_as_single_tu picks the winner's evaluation up from the overlays
  HM_SYNC();
  return bestMode;
}

// ------------------------------------------------------------------------------------------------
// chroma of a 16x16 CU whose luma transform is not split: one 8x8 block per component under the five chroma modes
// (estIntraPredChromaQT :2698-2849), ten evaluations from the CU's entry snapshot: sample work job after job on the 64 lanes, the RDOQ
// chains one job per lane (job = 2 * modeIndex + component - 1), then one lane per mode counts the mode's bits (chroma prediction mode,
// both cbfs, Cb and Cr coefficients: xGetIntraBitsQT :1038) and the modes are compared in the reference's order.  An 8x8 chroma block
// never tries transform skip and always scans diagonally.  Writes the winner (dirC / cbf / ts of the CU, coefficients, reconstruction
// into ws->reco) and returns its distortion.
// ------------------------------------------------------------------------------------------------
HM_FINL int s8_pred_sample_chroma(const Shared *e, int rs, int mode, int x, int y, int dcVal)
{ // TComPrediction::predIntraAng for an 8x8 chroma block (no edge filters, unfiltered reference lines of slot rs)
  const Pel *top = e->u.ref.refTop[rs], *left = e->u.ref.refLeft[rs];
  if (mode == PLANAR_IDX) {
    const int hor = (left[y + 1] << 3) + 8 + (x + 1) * (top[9] - left[y + 1]);
    const int ver = (top[x + 1] << 3) + (y + 1) * (left[9] - top[x + 1]);
    return (hor + ver) >> 4;
  }
  if (mode == DC_IDX) return dcVal;
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER_IDX : -(mode - HOR_IDX);
  const int absAng = HM_ANG_TABLE[hm_abs(angMode)], invAngle = HM_INV_ANG_TABLE[hm_abs(angMode)];
  const int angle = angMode < 0 ? -absAng : absAng;
  const Pel *mainR = isVer ? top : left, *sideR = isVer ? left : top;
  const int xx = isVer ? x : y, yy = isVer ? y : x;
  if (angle == 0) return mainR[xx + 1];
  const int deltaPos = (yy + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
  const int i0 = xx + di + 1;
  const int a = i0 >= 0 ? mainR[i0] : sideR[(128 - i0 * invAngle) >> 8];
  if (!df) return a;
  const int i1 = i0 + 1;
  const int b = i1 >= 0 ? mainR[i1] : sideR[(128 - i1 * invAngle) >> 8];
  return ((32 - df) * a + df * b + 16) >> 5;
}
HM_DEV HM_NOINLINE uint32_t simt8_chroma_cu16(Shared *e, int cuZ)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ);
  CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const TU t = tu_root(e, cuZ, 2);
  const int r = hm_z2r(cuZ), bitDepth = e->bitDepth;
  const int x4 = e->ctuX * 16 + (r & 15), y4 = e->ctuY * 16 + (r >> 4), px = e->ctuX * 32 + t.cx, py = e->ctuY * 32 + t.cy;
  init_adi_pattern(e, 2, px, py, 8, x4, y4, 4, 0);
  HM_PAR_FOR(i, 17) { e->u.ref.refTop[1][i] = e->u.ref.refTop[0][i]; e->u.ref.refLeft[1][i] = e->u.ref.refLeft[0][i]; }
  HM_SYNC();
  init_adi_pattern(e, 1, px, py, 8, x4, y4, 4, 0);
  int modeList[5] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX, DM_CHROMA_IDX};       // getAllowedChromaDir, TComDataCU.cpp:1486
  const int lumaDir = m->dirL[cuZ];
  for (int i = 0; i < 4; i++) if (lumaDir == modeList[i]) { modeList[i] = 34; break; }
  Simt8A *A = s8_a(e); Simt8B *B = s8_b(e);
  s8_setup(e, &e->cur, 10, 1, 5, 5, C_CHROMA_PRED);                  // chroma cbf at the CU's root TU: context 5 + transform depth 0
  const S8Par p = s8_params(e, 1);
  const uint32_t baseFrac = (uint32_t)(e->cur.frac & 32767);
  int dcVal[2];
  for (int c = 0; c < 2; c++) { int s = 0; HM_PAR_FOR(i, 8) s += e->u.ref.refTop[c][i + 1] + e->u.ref.refLeft[c][i + 1]; dcVal[c] = (hm_wave_sum_i(s) + 8) / 16; }
  const int s1 = 3 + bitDepth + 6 - 15, a1 = 1 << (s1 - 1), s2 = 9, a2 = 256;
  const int maxv = (1 << bitDepth) - 1, shiftSse = (bitDepth - 8) << 1, is2 = 20 - bitDepth;
  // ---- residual + forward transform of every job
  for (int job = 0; job < 10; job++) {
    const int c = job & 1, dirC = modeList[job >> 1], mode = dirC == DM_CHROMA_IDX ? lumaDir : dirC, ps = e->stride[1 + c];
    const Pel *org = e->fb.org[1 + c] + (size_t)py * ps + px;
    HM_PAR_FOR(l, 64) { const int y = l >> 3, x = l & 7; A->tile[0][l] = org[y * ps + x] - s8_pred_sample_chroma(e, c, mode, x, y, dcVal[c]); }
    HM_SYNC();
    HM_PAR_FOR(l, 64) {
      const int j = l >> 3, kk = l & 7; int32_t acc = 0;
      for (int i = 0; i < 8; i++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[0][j * 8 + i];
      A->tile[1][kk * 8 + j] = (acc + a1) >> s1;
    }
    HM_SYNC();
    HM_PAR_FOR(l, 64) {
      const int blkPos = B->scan[SCAN_DIAG][l], kk = blkPos >> 3, j = blkPos & 7; int32_t acc = 0;
      for (int i = 0; i < 8; i++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[1][j * 8 + i];
      A->cs[l][job] = (int16_t)((acc + a2) >> s2);
    }
    HM_SYNC();
  }
  HM_WAVE_FOR(k) { if (k < 10) A->outCbf[k] = (uint8_t)(s8_rdoq(A, B, p, k, SCAN_DIAG) > 0); }
  HM_SYNC();
  // ---- reconstruction + distortion of a job (also used for the winner's reconstruction afterwards)
  for (int pass = 0; pass < 12; pass++) {
    int job = pass;
    if (pass == 10) {                                                // all costs are in: bits and decision, then the winner's two jobs once more
      HM_WAVE_FOR(k) {
        if (k < 10 && !(k & 1)) {
          const int dirC = modeList[k >> 1];
          uint32_t frac = baseFrac;
          s8_bin(e, A, k, &frac, S8_MODE, dirC != DM_CHROMA_IDX);    // codeIntraDirChroma :692
          if (dirC != DM_CHROMA_IDX) frac += 2u * 32768u;
          const int cbfU = A->outCbf[k], cbfV = A->outCbf[k + 1];
          s8_bin(e, A, k, &frac, S8_CBF, cbfU);                      // xEncSubdivCbfQT :856: both cbfs at the root TU
          s8_bin(e, A, k, &frac, S8_CBF, cbfV);
          if (cbfU) s8_code_coeff(e, A, B, k, k, 1, SCAN_DIAG, &frac);
          if (cbfV) s8_code_coeff(e, A, B, k, k + 1, 1, SCAN_DIAG, &frac);
          const uint32_t dist = A->outDist[k] + A->outDist[k + 1];
          A->outCost[k] = calc_rd_cost(e, frac >> 15, dist);
        }
      }
      HM_SYNC();
      double bestCost = HM_MAX_DOUBLE; int best = 0;
      for (int mi = 0; mi < 5; mi++) { const double v = A->outCost[2 * mi]; if (v < bestCost) { bestCost = v; best = mi; } }
      best = HM_UNI(best);
      if (hm_lane() == 0) A->pad[0] = (uint8_t)best;
      HM_SYNC();
    }
    const int best = pass >= 10 ? A->pad[0] : 0;
    if (pass >= 10) job = 2 * best + (pass - 10);
    const int c = job & 1, dirC = modeList[job >> 1], mode = dirC == DM_CHROMA_IDX ? lumaDir : dirC, ps = e->stride[1 + c], cbf = A->outCbf[job];
    const Pel *org = e->fb.org[1 + c] + (size_t)py * ps + px;
    if (cbf) {
      HM_PAR_FOR(l, 64) {
        const int lvv = hm_clip3(p.dqMin, p.dqMax, B->dc[l][job] >> 16);
        int v;
        if (p.dqShift > 0) v = (lvv * p.dqScale + (1 << (p.dqShift - 1))) >> p.dqShift;
        else v = (int)((unsigned)(lvv * p.dqScale) << (-p.dqShift));
        A->tile[0][B->scan[SCAN_DIAG][l]] = hm_clip3(-32768, 32767, v);
      }
      HM_SYNC();
      HM_PAR_FOR(l, 64) {
        const int j = l >> 3, i = l & 7; int32_t acc = 0;
        for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[0][kk * 8 + j];
        A->tile[1][j * 8 + i] = hm_clip3(-32768, 32767, (acc + 64) >> 7);
      }
      HM_SYNC();
    }
    uint32_t sse = 0;
    const int po = HM_PLANE_OFF(1 + c);
    HM_PAR_FOR(l, 64) {
      const int j = l >> 3, i = l & 7; int resi = 0;
      if (cbf) {
        int32_t acc = 0;
        for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[1][kk * 8 + j];
        resi = hm_clip3(-32768, 32767, (acc + (1 << (is2 - 1))) >> is2);
      }
      const int rr = hm_clip3(0, maxv, s8_pred_sample_chroma(e, c, mode, i, j, dcVal[c]) + resi);
      const int d = org[j * ps + i] - rr; sse += (uint32_t)((d * d) >> shiftSse);
      if (pass >= 10) {                                              // the winner: reconstruction and levels to where the CU's data lives
        ws->reco[po + (t.cy + j) * 32 + t.cx + i] = (Pel)rr;
        e->cc[po + t.cOff + B->scan[SCAN_DIAG][l]] = cbf ? (B->dc[l][job] >> 16) : 0;
      }
    }
    const uint32_t dsum = hm_wave_sum(sse);
    if (pass < 10 && hm_lane() == 0) A->outDist[job] = (uint32_t)(e->fb.chromaWeight * (double)dsum);   // getDistPart, TComRdCost.cpp:447-450
    HM_SYNC();
  }
  const int best = A->pad[0], cbfU = A->outCbf[2 * best], cbfV = A->outCbf[2 * best + 1];
  const uint32_t bestDist = A->outDist[2 * best] + A->outDist[2 * best + 1];
  const int bm = modeList[best];
  HM_PAR_FOR(i, 16) { m->cbf[1][cuZ + i] = (uint8_t)cbfU; m->cbf[2][cuZ + i] = (uint8_t)cbfV; m->ts[1][cuZ + i] = 0; m->ts[2][cuZ + i] = 0; m->dirC[cuZ + i] = (uint8_t)bm; }
  HM_SYNC();
  return bestDist;
}

// The closing pass of estIntraPredQT (:2566-2600) starts with the unsplit evaluation of the winner's 8x8 transform block -- the very
// evaluation the first pass above made for it, from the same snapshot.  Instead of repeating it (xIntraCodingTUBlock + xGetIntraBitsQT), the
// winner's results are put where the residual quadtree expects them: levels and reconstruction in the layer buffers of the 8x8 size, the
// estimator (e->cur) advanced past the block's syntax (the bins every candidate codes alike, then the contexts and the bit count of the
// winner's lane).  Must run right after simt8_luma_first_pass (the overlays and the reference lines are still in place), with e->cur
// holding the CU's entry snapshot.  Distortion / bits / cbf in e->outDistY / e->outBits / e->outDist.
HM_DEV HM_NOINLINE void simt8_luma_winner_as_single_tu(Shared *e, TU tv)
{
  HM_ENTRY(e); tv = hm_uni_struct(tv);
  const TU *t = &tv; WorkSpace *ws = e->ws;
  Simt8A *A = s8_a(e); Simt8B *B = s8_b(e);
  const int best = HM_UNI(e->s8Winner), z = t->cuZ + t->relZ, ps = e->stride[0], bitDepth = e->bitDepth;
  const int mode = e->rdModeList[best], cbf = A->outCbf[best];
  int scanType = SCAN_DIAG;
  if (hm_abs(mode - VER_IDX) <= 4) scanType = SCAN_HOR; else if (hm_abs(mode - HOR_IDX) <= 4) scanType = SCAN_VER;
  const S8Par p = s8_params(e, 0);
  int dcVal;
  { int s = 0; HM_PAR_FOR(i, 8) s += e->u.ref.refTop[0][i + 1] + e->u.ref.refLeft[0][i + 1]; dcVal = (hm_wave_sum_i(s) + 8) / 16; }
  const int maxv = (1 << bitDepth) - 1, is2 = 20 - bitDepth;
  TCoeff *coef = ws->qtCoef[2] + z * 16; Pel *rq = ws->qtRec[2] + t->y * 64 + t->x;
  Pel *recPic = e->fb.rec[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x;
  if (cbf) {
    HM_PAR_FOR(l, 64) {
      const int lv = B->dc[l][best] >> 16, blk = B->scan[scanType][l];
      coef[blk] = lv;
      const int lvv = hm_clip3(p.dqMin, p.dqMax, lv);
      int v;
      if (p.dqShift > 0) v = (lvv * p.dqScale + (1 << (p.dqShift - 1))) >> p.dqShift;
      else v = (int)((unsigned)(lvv * p.dqScale) << (-p.dqShift));
      A->tile[0][blk] = hm_clip3(-32768, 32767, v);
    }
    HM_SYNC();
    HM_PAR_FOR(l, 64) {
      const int j = l >> 3, i = l & 7; int32_t acc = 0;
      for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[0][kk * 8 + j];
      A->tile[1][j * 8 + i] = hm_clip3(-32768, 32767, (acc + 64) >> 7);
    }
    HM_SYNC();
  } else { HM_PAR_FOR(l, 64) coef[l] = 0; }
  HM_PAR_FOR(l, 64) {
    const int j = l >> 3, i = l & 7; int resi = 0;
    if (cbf) {
      int32_t acc = 0;
      for (int kk = 0; kk < 8; kk++) acc += HM_LT()->tmat[(kk * 4) * HM_TSTRIDE + i] * A->tile[1][kk * 8 + j];
      resi = hm_clip3(-32768, 32767, (acc + (1 << (is2 - 1))) >> is2);
    }
    const Pel rr = (Pel)hm_clip3(0, maxv, pred_sample(e, mode, 8, 3, i, j, dcVal, bitDepth) + resi);
    rq[j * 64 + i] = rr; recPic[j * ps + i] = rr;
  }
  HM_SYNC();
  { // the estimator: the bins in front of the lane's own (same as simt8_luma_first_pass counted), then the lane's contexts and bit count
    CabacR r; cabr_load(e, r, &e->cur);
    r.frac &= 32767;
    if (e->im) { code_skip_flag(e, &r, t->cuZ); enc_bin(e, &r, C_PRED_MODE, 1); }
    enc_bin(e, &r, C_PART, 1);
    enc_bin(e, &r, C_SUBDIV + 2, 0);
    cabr_store(r, &e->cur);
  }
  HM_PAR_FOR(j, S8_NCTX - 1) {
    int c;
    if (j < S8_ONE) c = C_SIG + j; else if (j < S8_ABS) c = C_ONE + (j - S8_ONE); else if (j < S8_LX) c = C_ABS + (j - S8_ABS);
    else if (j < S8_LY) c = C_LASTX + 3 + (j - S8_LX); else if (j < S8_CG) c = C_LASTY + 3 + (j - S8_LY);
    else if (j < S8_CBF) c = C_SIG_CG + (j - S8_CG); else if (j == S8_CBF) c = C_QT_CBF + 1; else c = C_INTRA_LUMA;
    e->cur.s[c] = A->ctx[j][best];
  }
  e->cur.frac = (uint64_t)A->outFrac[best];
  e->outDistY = A->outDist[best]; e->outBits = A->outFrac[best] >> 15; e->outDist = (uint32_t)cbf;
  HM_SYNC();
}
