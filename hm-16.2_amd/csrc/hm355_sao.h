// hm355 -- sample adaptive offset, encoder side (TEncSampleAdaptiveOffset::SAOProcess, TLibEncoder/TEncSampleAdaptiveOffset.cpp:259;
// SURVEY.md section 8f, n1), on the deblocked picture of a slot:
//   hm355_sao_stats_kernel   per-CTU statistics of every SAO type (getBlkStats :910), one workgroup per (CTU, component), lanes over
//                            the samples, class histograms in LDS
//   hm355_sao_decide_kernel  picture-level on/off (decidePicParams :365) and the per-CTU off / new / merge decision in CTU order
//                            (decideBlkParams :780): the CABAC estimator state chains from CTU to CTU, so one lane walks the picture
//                            while the wavefront stages its inputs and outputs through LDS; pictures of a batch run side by side
//   hm355_sao_apply_kernel   offsets applied per sample (TComSampleAdaptiveOffset::offsetBlock, TComSampleAdaptiveOffset.cpp:311)
// One slice, no tiles, SAOLcuBoundary = 0, offset bit shifts 0 (every cfg of the reference).
#pragma once

enum { SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, SAO_EO_0 = 0, SAO_EO_90, SAO_EO_135, SAO_EO_45, SAO_BO, SAO_NUM_TYPES };
struct SaoOff { int32_t mode, type, aux, offset[32]; };        // SAOOffset (TypeDef.h)
struct SaoBlk { SaoOff c[3]; };                                 // SAOBlkParam
struct SaoStat { int32_t diff[SAO_NUM_TYPES][32], count[SAO_NUM_TYPES][32]; };    // SAOStatData of the five types of one (CTU, component)
struct SaoCand { int32_t aux; int32_t offset[32]; int64_t dist; };   // best offsets of one (CTU, component, type) and their distortion: independent of the CABAC state
struct SaoParams {                                             // one picture of the batch
  int32_t qp, cabacInitType, depth, enabled[3], numOff[3];
  double lambda[3], disabledPrev[3];                           // m_lambda; m_saoDisabledRate[comp][depth - 1]
  SaoStat *stat;                                               // [numCtus][3]
  SaoCand *cand;                                               // [numCtus][3][SAO_NUM_TYPES]
  SaoBlk *coded, *recon;                                       // [numCtus] parameters as coded / with reconstructed offsets
  Pel *src[3];                                                 // copy of the deblocked planes (same padded layout as FrameBuf::rec)
};
HM_CONST int8_t HM_SAO_DX[4][2] = { {-1, 1}, {0, 0}, {-1, 1}, {1, -1} };
HM_CONST int8_t HM_SAO_DY[4][2] = { {0, 0}, {-1, 1}, {-1, 1}, {-1, 1} };
HM_DEV inline int sao_sgn(int v) { return (v > 0) - (v < 0); }

#if !defined(HM355_HOSTSIM)
__device__ inline int64_t sao_est_dist(int64_t count, int64_t offset, int64_t diffSum, int shift) { return (count * offset * offset - diffSum * offset * 2) >> shift; }
__device__ int64_t sao_distortion(int bd, int type, int aux, const int32_t *invQuantOffset, const int32_t *diff, const int32_t *count)
{ // getDistortion :399
  const int shift = 2 * (bd - 8); int64_t dist = 0;
  if (type != SAO_BO) { for (int i = 0; i < 5; i++) dist += sao_est_dist(count[i], invQuantOffset[i], diff[i], shift); }
  else for (int i = aux; i < aux + 4; i++) { const int b = i % 32; dist += sao_est_dist(count[b], invQuantOffset[b], diff[b], shift); }
  return dist;
}
__device__ int sao_est_iter_offset(int type, double lambda, int offsetInput, int64_t count, int64_t diffSum, int shift, int64_t *bestDist, double *bestCost, int offsetTh)
{ // estIterOffset :443
  int iterOffset = offsetInput, offsetOutput = 0;
  double tempMinCost = lambda;
  while (iterOffset != 0) {
    int64_t tempRate = (type == SAO_BO) ? (hm_abs(iterOffset) + 2) : (hm_abs(iterOffset) + 1);
    if (hm_abs(iterOffset) == offsetTh) tempRate--;
    const int64_t tempDist = sao_est_dist(count, iterOffset, diffSum, shift);
    const double tempCost = (double)tempDist + lambda * (double)tempRate;
    if (tempCost < tempMinCost) { tempMinCost = tempCost; offsetOutput = iterOffset; *bestDist = tempDist; *bestCost = tempCost; }
    iterOffset = (iterOffset > 0) ? (iterOffset - 1) : (iterOffset + 1);
  }
  return offsetOutput;
}
__device__ inline double sao_round_ibdi(int bitDepth, double x)
{ // xRoundIbdi / xRoundIbdi2 :51-59
  if (bitDepth > 8) return (x > 0) ? (int)(((int)x + (1 << (bitDepth - 8 - 1))) / (1 << (bitDepth - 8))) : (int)(((int)x - (1 << (bitDepth - 8 - 1))) / (1 << (bitDepth - 8)));
  return x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
}
__device__ void sao_derive_offsets(int bd, int offsetTh, double lambda, int type, const int32_t *diff, const int32_t *count, int32_t *quantOffsets, int32_t *aux)
{ // deriveOffsets :476-580
  const int shift = 2 * (bd - 8);
  for (int k = 0; k < 32; k++) quantOffsets[k] = 0;
  const int numClasses = type == SAO_BO ? 32 : 5;
  for (int k = 0; k < numClasses; k++) {
    if (type != SAO_BO && k == 2) continue;
    if (count[k] == 0) continue;
    int q = (int)sao_round_ibdi(bd, (double)((int64_t)diff[k] << (bd - 8)) / (double)((int64_t)count[k]));
    quantOffsets[k] = hm_clip3(-offsetTh, offsetTh, q);
  }
  if (type != SAO_BO) {
    int64_t classDist; double classCost;
    for (int k = 0; k < 5; k++) {
      if ((k == 0 || k == 1) && quantOffsets[k] < 0) quantOffsets[k] = 0;
      if ((k == 3 || k == 4) && quantOffsets[k] > 0) quantOffsets[k] = 0;
      if (quantOffsets[k] != 0) quantOffsets[k] = sao_est_iter_offset(type, lambda, quantOffsets[k], count[k], diff[k], shift, &classDist, &classCost, offsetTh);
    }
    *aux = 0;
  } else {
    double costBO[32];
    for (int k = 0; k < 32; k++) {
      int64_t distK = 0; costBO[k] = lambda;
      if (quantOffsets[k] != 0) quantOffsets[k] = sao_est_iter_offset(type, lambda, quantOffsets[k], count[k], diff[k], shift, &distK, &costBO[k], offsetTh);
    }
    double minCost = HM_MAX_DOUBLE;
    for (int band = 0; band < 32 - 4 + 1; band++) {
      double cost = costBO[band]; cost += costBO[band + 1]; cost += costBO[band + 2]; cost += costBO[band + 3];
      if (cost < minCost) { minCost = cost; *aux = band; }
    }
    for (int k = 0; k < 32; k++) { const int rel = (k - *aux + 32) % 32; if (rel >= 4) quantOffsets[k] = 0; }
  }
}
// grid: x = CTU, y = component, z = picture
extern "C" __global__ void __launch_bounds__(64) hm355_sao_stats_kernel(const Params *P, SaoParams *sps)
{
  __shared__ int32_t hist[2][SAO_NUM_TYPES][32];
  const FrameBuf *fb = P->frames + blockIdx.z; SaoParams *sp = sps + blockIdx.z;
  const int a = (int)blockIdx.x, comp = (int)blockIdx.y, lane = (int)threadIdx.x;
  for (int i = lane; i < 2 * SAO_NUM_TYPES * 32; i += 64) (&hist[0][0][0])[i] = 0;
  __syncthreads();
  const int sh = comp ? 1 : 0, xPos = (a % P->wCtu) * 64, yPos = (a / P->wCtu) * 64;
  const int hFull = (yPos + 64 > P->height) ? P->height - yPos : 64, wFull = (xPos + 64 > P->width) ? P->width - xPos : 64;
  const int w = wFull >> sh, h = hFull >> sh, bx = xPos >> sh, by = yPos >> sh, st = P->stride[comp];
  const int L = xPos > 0, A = yPos > 0, R = xPos + 64 < P->width, B = yPos + 64 < P->height;
  const int skipR = comp ? 3 : 5, skipB = comp ? 2 : 4;                 // m_skipLinesR / m_skipLinesB :136-140
  // sample ranges of the statistics per type (getBlkStats with isCalculatePreDeblockSamples == false)
  const int x0e = L ? 0 : 1, x1e = R ? w - skipR : w - 1, x1f = R ? w - skipR : w;
  const int y0e = A ? 0 : 1, y1e = B ? h - skipB : h - 1, y1f = B ? h - skipB : h;
  const Pel *src = sp->src[comp], *org = fb->org[comp];
  const int bdShift = P->bitDepth - 5;
  for (int i = lane; i < w * h; i += 64) {
    const int y = i / w, x = i - y * w, px = bx + x, py = by + y;
    const Pel *p = src + (size_t)py * st + px;
    const int c = p[0], d = (int)org[(size_t)py * st + px] - c;
    const int inX = x >= x0e && x < x1e, inY = y >= y0e && y < y1e;
    if (inX && y < y1f) { const int k = 2 + sao_sgn(c - p[-1]) + sao_sgn(c - p[1]); atomicAdd(&hist[0][SAO_EO_0][k], d); atomicAdd(&hist[1][SAO_EO_0][k], 1); }
    if (x < x1f && inY) { const int k = 2 + sao_sgn(c - p[-st]) + sao_sgn(c - p[st]); atomicAdd(&hist[0][SAO_EO_90][k], d); atomicAdd(&hist[1][SAO_EO_90][k], 1); }
    if (inX && inY) {
      int k = 2 + sao_sgn(c - p[-st - 1]) + sao_sgn(c - p[st + 1]); atomicAdd(&hist[0][SAO_EO_135][k], d); atomicAdd(&hist[1][SAO_EO_135][k], 1);
      k = 2 + sao_sgn(c - p[-st + 1]) + sao_sgn(c - p[st - 1]); atomicAdd(&hist[0][SAO_EO_45][k], d); atomicAdd(&hist[1][SAO_EO_45][k], 1);
    }
    if (x < x1f && y < y1f) { const int k = c >> bdShift; atomicAdd(&hist[0][SAO_BO][k], d); atomicAdd(&hist[1][SAO_BO][k], 1); }
  }
  __syncthreads();
  SaoStat *out = sp->stat + (size_t)a * 3 + comp;
  for (int i = lane; i < SAO_NUM_TYPES * 32; i += 64) { (&out->diff[0][0])[i] = (&hist[0][0][0])[i]; (&out->count[0][0])[i] = (&hist[1][0][0])[i]; }
  // the "new offsets" candidate of every type (deriveOffsets :476, getDistortion :399) depends only on these statistics and lambda,
  // not on the CABAC state, so it is prepared here, one lane per type, and the serial decision only counts bits and compares
  if (lane < SAO_NUM_TYPES) {
    const int type = lane, bd = P->bitDepth, maxOffQ = (1 << ((bd < 10 ? bd : 10) - 5)) - 1;
    SaoCand *cd = sp->cand + ((size_t)a * 3 + comp) * SAO_NUM_TYPES + type;
    int32_t off[32], aux = 0;
    sao_derive_offsets(bd, maxOffQ, sp->lambda[comp], type, hist[0][type], hist[1][type], off, &aux);
    cd->aux = aux; cd->dist = sao_distortion(bd, type, aux, off, hist[0][type], hist[1][type]);
    for (int k = 0; k < 32; k++) cd->offset[k] = off[k];
  }
}

// ---- decision (one lane per picture) ----
struct SaoCab { uint8_t s[2]; uint64_t frac; };               // ctx 0 = sao_merge_*_flag, ctx 1 = sao_type_idx_*
__device__ inline void sao_bin(SaoCab *c, int ctx, int bin) { const int st = c->s[ctx]; c->frac += (uint64_t)HM_ENTROPY_BITS[st ^ bin]; c->s[ctx] = (uint8_t)hm_next_state(st, bin); }
__device__ inline void sao_ep(SaoCab *c, int n) { c->frac += (uint64_t)32768 * (uint64_t)n; }
__device__ inline uint32_t sao_bits(const SaoCab *c) { return (uint32_t)(c->frac >> 15); }
__device__ inline void sao_reset_bits(SaoCab *c) { c->frac &= 32767; }
__device__ inline void sao_code_max_uvlc(SaoCab *c, int code, int maxSymbol)
{ if (maxSymbol == 0) return; if (code == 0) { sao_ep(c, 1); return; } sao_ep(c, 1 + (code - 1) + (maxSymbol > code ? 1 : 0)); }
__device__ void sao_code_offset_param(SaoCab *c, int comp, const SaoOff *p, int sliceEnabled, int maxOffQ)
{ // TEncSbac::codeSAOOffsetParam, TEncSbac.cpp:1597
  if (!sliceEnabled) return;
  const int first = comp != 2;
  if (first) { const int sym = p->mode == SAO_OFF ? 0 : (p->type == SAO_BO ? 1 : 2); if (sym == 0) sao_bin(c, 1, 0); else { sao_bin(c, 1, 1); sao_ep(c, 1); } }
  if (p->mode == SAO_NEW) {
    int offset[4], k = 0;
    const int numClasses = p->type == SAO_BO ? 4 : 5;
    for (int i = 0; i < numClasses; i++) { if (p->type != SAO_BO && i == 2) continue; offset[k++] = p->offset[p->type == SAO_BO ? (p->aux + i) % 32 : i]; }
    for (int i = 0; i < 4; i++) sao_code_max_uvlc(c, hm_abs(offset[i]), maxOffQ);
    if (p->type == SAO_BO) { for (int i = 0; i < 4; i++) if (offset[i] != 0) sao_ep(c, 1); sao_ep(c, 5); }
    else if (first) sao_ep(c, 2);
  }
}
__device__ void sao_code_blk_param(SaoCab *c, const SaoBlk *p, const int *sliceEnabled, int leftAvail, int aboveAvail, int onlyMergeInfo, int maxOffQ)
{ // TEncSbac::codeSAOBlkParam :1674
  int isLeft = 0, isAbove = 0;
  if (leftAvail) { isLeft = p->c[0].mode == SAO_MERGE && p->c[0].type == 0; sao_bin(c, 0, isLeft); }
  if (aboveAvail && !isLeft) { isAbove = p->c[0].mode == SAO_MERGE && p->c[0].type == 1; sao_bin(c, 0, isAbove); }
  if (onlyMergeInfo) return;
  if (!isLeft && !isAbove) for (int comp = 0; comp < 3; comp++) sao_code_offset_param(c, comp, &p->c[comp], sliceEnabled[comp], maxOffQ);
}
// One workgroup per picture.  The decision itself is serial (lane 0: the estimator state and the merge candidates chain from CTU to
// CTU), but everything it reads and writes for a CTU is moved between HBM and LDS by the whole wavefront, so the serial part never waits
// on a dependent HBM access: candidates of the 15 (component, type) pairs, the reconstructed parameters of the left / above CTU, the
// statistics of this CTU for the neighbours' types (merge distortion), and the coded / reconstructed parameters going out.
extern "C" __global__ void __launch_bounds__(64) hm355_sao_decide_kernel(const Params *P, SaoParams *sps)
{
  __shared__ SaoCand sCand[3 * SAO_NUM_TYPES];
  __shared__ SaoBlk sNb[2], sMode, sCoded, sRecon;
  __shared__ int32_t sMrg[2][3][2][32];                           // [merge candidate][component][diff, count][class] of the neighbour's type
  SaoParams *sp = sps + blockIdx.x;
  const int lane = (int)threadIdx.x;
  const int bd = P->bitDepth, numCtus = P->wCtu * P->hCtu, wCtu = P->wCtu;
  const int maxOffQ = (1 << ((bd < 10 ? bd : 10) - 5)) - 1;                                   // g_saoMaxOffsetQVal
  int sliceEnabled[3];                                                                      // decidePicParams :365
  for (int c = 0; c < 3; c++) sliceEnabled[c] = !(sp->depth > 0 && sp->disabledPrev[c] > (c == 0 ? 0.75 : 0.5));
  int numOff[3] = {0, 0, 0};
  SaoCab cur, cabCur, cabNext, cabMid, cabTemp;
  { // initRDOCabacCoder :247 (INIT_SAO_MERGE_FLAG / INIT_SAO_TYPE_IDX, ContextTables.h:445-458: rows B, P, I)
    const int qp = hm_clip3(0, 51, sp->qp);
    for (int i = 0; i < 2; i++) {
      const int iv = i == 0 ? 153 : (sp->cabacInitType == 0 ? 160 : (sp->cabacInitType == 1 ? 185 : 200));
      const int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
      int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
      const int mps = st >= 64;
      cur.s[i] = (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
    }
    cur.frac = 0;
  }
  cabNext = cabTemp = cur;
  const int allDisabled = !sliceEnabled[0] && !sliceEnabled[1] && !sliceEnabled[2];
  const int nW = (int)(sizeof(SaoBlk) / 4), nC = (int)(sizeof(sCand) / 4);
  for (int a = 0; a < numCtus; a++) {                                                        // decideBlkParams :780-860
    const int hasL = (a % wCtu) > 0, hasA = (a / wCtu) > 0;
    if (!allDisabled) {
      const int32_t *gc = (const int32_t *)(sp->cand + (size_t)a * 3 * SAO_NUM_TYPES);
      for (int i = lane; i < nC; i += 64) ((int32_t *)sCand)[i] = gc[i];
      if (hasL) { const int32_t *g = (const int32_t *)(sp->recon + a - 1); for (int i = lane; i < nW; i += 64) ((int32_t *)&sNb[0])[i] = g[i]; }
      if (hasA) { const int32_t *g = (const int32_t *)(sp->recon + a - wCtu); for (int i = lane; i < nW; i += 64) ((int32_t *)&sNb[1])[i] = g[i]; }
      __syncthreads();
      for (int i = lane; i < 2 * 3 * 64; i += 64) {                                             // this CTU's statistics for the neighbours' types
        const int mt = i / 192, r = i - mt * 192, comp = r >> 6, k = r & 63;
        if (mt == 0 ? hasL : hasA) {
          const SaoOff *m = &sNb[mt].c[comp];
          if (m->mode != SAO_OFF) { const SaoStat *sd = sp->stat + (size_t)a * 3 + comp; sMrg[mt][comp][k >> 5][k & 31] = k < 32 ? sd->diff[m->type][k] : sd->count[m->type][k - 32]; }
        }
      }
      __syncthreads();
    }
    if (lane == 0) {
      SaoBlk *coded = &sCoded, *recon = &sRecon, *mode = &sMode;
      if (allDisabled) { for (int c = 0; c < 3; c++) { coded->c[c].mode = SAO_OFF; coded->c[c].type = coded->c[c].aux = 0; for (int k = 0; k < 32; k++) coded->c[c].offset[k] = 0; recon->c[c] = coded->c[c]; } }
      else {
        cabCur = cur;
        double minCost = HM_MAX_DOUBLE, modeCost;
        { // deriveModeNewRDO :583-723
          double mc, cost; uint32_t prevBits; int64_t dist[3], modeDist[3] = {0, 0, 0};
          SaoOff test[3];
          for (int c = 0; c < 3; c++) { mode->c[c].mode = SAO_OFF; mode->c[c].type = mode->c[c].aux = 0; for (int k = 0; k < 32; k++) mode->c[c].offset[k] = 0; test[c] = mode->c[c]; }
          cur = cabCur;
          sao_code_blk_param(&cur, mode, sliceEnabled, hasL, hasA, 1, maxOffQ);
          cabMid = cur;
          { const int comp = 0;
            sao_reset_bits(&cur);
            sao_code_offset_param(&cur, comp, &mode->c[comp], sliceEnabled[comp], maxOffQ);
            mc = sp->lambda[comp] * (double)sao_bits(&cur);
            cabTemp = cur;
            if (sliceEnabled[comp]) for (int type = 0; type < SAO_NUM_TYPES; type++) {
              const SaoCand *cd = &sCand[comp * SAO_NUM_TYPES + type];                              // deriveOffsets + getDistortion, prepared by the statistics kernel
              test[comp].mode = SAO_NEW; test[comp].type = type; test[comp].aux = cd->aux; for (int k = 0; k < 32; k++) test[comp].offset[k] = cd->offset[k];
              dist[comp] = cd->dist;
              cur = cabMid; sao_reset_bits(&cur);
              sao_code_offset_param(&cur, comp, &test[comp], sliceEnabled[comp], maxOffQ);
              cost = (double)dist[comp] + sp->lambda[comp] * (double)(int)sao_bits(&cur);
              if (cost < mc) { mc = cost; modeDist[comp] = dist[comp]; mode->c[comp] = test[comp]; cabTemp = cur; }
            }
            cur = cabTemp; cabMid = cur;
          }
          cost = 0; prevBits = 0; sao_reset_bits(&cur);
          for (int comp = 1; comp < 3; comp++) {
            sao_code_offset_param(&cur, comp, &mode->c[comp], sliceEnabled[comp], maxOffQ);
            const uint32_t cw = sao_bits(&cur); cost += sp->lambda[comp] * (cw - prevBits); prevBits = cw;
          }
          mc = cost;
          for (int type = 0; type < SAO_NUM_TYPES; type++) {
            cur = cabMid; sao_reset_bits(&cur); prevBits = 0; cost = 0;
            for (int comp = 1; comp < 3; comp++) {
              if (!sliceEnabled[comp]) { test[comp].mode = SAO_OFF; dist[comp] = 0; continue; }
              const SaoCand *cd = &sCand[comp * SAO_NUM_TYPES + type];
              test[comp].mode = SAO_NEW; test[comp].type = type; test[comp].aux = cd->aux; for (int k = 0; k < 32; k++) test[comp].offset[k] = cd->offset[k];
              dist[comp] = cd->dist;
              sao_code_offset_param(&cur, comp, &test[comp], sliceEnabled[comp], maxOffQ);
              const uint32_t cw = sao_bits(&cur); cost += dist[comp] + (sp->lambda[comp] * (cw - prevBits)); prevBits = cw;
            }
            if (cost < mc) { mc = cost; for (int comp = 1; comp < 3; comp++) { modeDist[comp] = dist[comp]; mode->c[comp] = test[comp]; } }
          }
          modeCost = 0;
          for (int comp = 0; comp < 3; comp++) modeCost += (double)modeDist[comp] / sp->lambda[comp];
          cur = cabCur; sao_reset_bits(&cur);
          sao_code_blk_param(&cur, mode, sliceEnabled, hasL, hasA, 0, maxOffQ);
          modeCost += (double)sao_bits(&cur);
        }
        if (modeCost < minCost) { minCost = modeCost; *coded = *mode; cabNext = cur; }
        { // deriveModeMergeRDO :726-777
          double best = HM_MAX_DOUBLE; int bestType = -1;
          for (int mt = 0; mt < 2; mt++) {
            if (!(mt == 0 ? hasL : hasA)) continue;
            double normDist = 0;
            for (int comp = 0; comp < 3; comp++) {
              const SaoOff *m = &sNb[mt].c[comp];
              if (m->mode != SAO_OFF) normDist += ((double)sao_distortion(bd, m->type, m->aux, m->offset, sMrg[mt][comp][0], sMrg[mt][comp][1])) / sp->lambda[comp];
            }
            SaoBlk *t = mode;                                                                  // only mode / type of component 0 matter to the syntax
            for (int comp = 0; comp < 3; comp++) { t->c[comp].mode = SAO_MERGE; t->c[comp].type = mt; }
            cur = cabCur; sao_reset_bits(&cur);
            sao_code_blk_param(&cur, t, sliceEnabled, hasL, hasA, 0, maxOffQ);
            const double cost = normDist + (double)(int)sao_bits(&cur);
            if (cost < best) { best = cost; bestType = mt; cabTemp = cur; }
          }
          cur = cabTemp;
          if (best < minCost) {
            minCost = best;
            for (int comp = 0; comp < 3; comp++) { coded->c[comp] = sNb[bestType].c[comp]; coded->c[comp].mode = SAO_MERGE; coded->c[comp].type = bestType; }
            cabNext = cur;
          }
        }
        cur = cabNext;
        // reconstructBlkSAOParam :248: new -> offsets as coded (zero bit shift), merge -> the neighbour's reconstructed parameters
        for (int comp = 0; comp < 3; comp++) { if (coded->c[comp].mode == SAO_MERGE) recon->c[comp] = sNb[coded->c[comp].type].c[comp]; else recon->c[comp] = coded->c[comp]; }
      }
      for (int c = 0; c < 3; c++) numOff[c] += recon->c[c].mode == SAO_OFF;                      // SAO_ENCODING_CHOICE :868
    }
    __syncthreads();
    { int32_t *gc = (int32_t *)(sp->coded + a), *gr = (int32_t *)(sp->recon + a);
      for (int i = lane; i < nW; i += 64) { gc[i] = ((const int32_t *)&sCoded)[i]; gr[i] = ((const int32_t *)&sRecon)[i]; } }
    __threadfence_block();
    __syncthreads();
  }
  if (lane == 0) for (int c = 0; c < 3; c++) { sp->enabled[c] = sliceEnabled[c]; sp->numOff[c] = numOff[c]; }
}

// grid: x = ceil(plane width / 64), y = plane row, z = 3 * picture + component
extern "C" __global__ void __launch_bounds__(64) hm355_sao_apply_kernel(const Params *P, SaoParams *sps)
{
  const int f = (int)blockIdx.z / 3, comp = (int)blockIdx.z % 3, sh = comp ? 1 : 0;
  const FrameBuf *fb = P->frames + f; const SaoParams *sp = sps + f;
  const int pw = P->width >> sh, ph = P->height >> sh, px = (int)(blockIdx.x * 64 + threadIdx.x), py = (int)blockIdx.y;
  if (px >= pw || py >= ph) return;
  const int a = (py >> (6 - sh)) * P->wCtu + (px >> (6 - sh));
  const SaoOff *o = &sp->recon[a].c[comp];
  if (o->mode == SAO_OFF) return;
  const int st = P->stride[comp], t = o->type;
  const Pel *p = sp->src[comp] + (size_t)py * st + px;
  const int c = p[0];
  int off;
  if (t == SAO_BO) off = o->offset[c >> (P->bitDepth - 5)];
  else {                                                               // an edge-offset sample changes when both neighbours lie inside the picture
    const int ax = px + HM_SAO_DX[t][0], ay = py + HM_SAO_DY[t][0], qx = px + HM_SAO_DX[t][1], qy = py + HM_SAO_DY[t][1];
    if (ax < 0 || ax >= pw || ay < 0 || ay >= ph || qx < 0 || qx >= pw || qy < 0 || qy >= ph) return;
    off = o->offset[2 + sao_sgn(c - p[HM_SAO_DY[t][0] * st + HM_SAO_DX[t][0]]) + sao_sgn(c - p[HM_SAO_DY[t][1] * st + HM_SAO_DX[t][1]])];
  }
  fb->rec[comp][(size_t)py * st + px] = (Pel)hm_clip3(0, (1 << P->bitDepth) - 1, c + off);
}
#endif
