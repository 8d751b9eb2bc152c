// Host-side helpers shared by the HIP library (hm355_capi.hip) and the host simulation used for
// debugging (tests/hostsim): lookup-table generation, slice parameter derivation, CTU scheduling.
#pragma once
#include "hm355_types.h"
#include <math.h>
#include <string.h>
#include <vector>

// z-scan <-> raster of the 16x16 grid of 4x4 partitions (TComRom.cpp:256-290) and the coefficient
// scans (ScanGenerator, TComRom.cpp:52-225)
static inline void hm355_gen_scan(int w, int h, int stride, int type, int offx, int offy, uint16_t *out, int count)
{
  int line = 0, col = 0;
  for (int i = 0; i < count; i++) {
    out[i] = (uint16_t)((line + offy) * stride + col + offx);
    if (type == 0) {
      if (col == w - 1 || line == 0) { line += col + 1; col = 0; if (line >= h) { col += line - (h - 1); line = h - 1; } }
      else { col++; line--; }
    } else if (type == 1) { if (col == w - 1) { line++; col = 0; } else col++; }
    else { if (line == h - 1) { col++; line = 0; } else line++; }
  }
}
static inline void hm355_build_tables(Tables *t)
{
  memset(t, 0, sizeof(*t));
  for (int z = 0; z < 256; z++) { // bit de-interleave: z-scan index -> (x,y)
    int x = 0, y = 0;
    for (int b = 0; b < 4; b++) { x |= ((z >> (2 * b)) & 1) << b; y |= ((z >> (2 * b + 1)) & 1) << b; }
    t->z2r[z] = (uint8_t)(y * 16 + x); t->r2z[y * 16 + x] = (uint8_t)z;
  }
  for (int ty = 0; ty < 3; ty++)
    for (int l = 0; l < 4; l++) {
      const int n = 4 << l, g = n >> 2;
      hm355_gen_scan(g, g, g, ty, 0, 0, t->scanCG[ty][l], g * g);
      for (int gi = 0; gi < g * g; gi++) {
        const int gx = t->scanCG[ty][l][gi] % g, gy = t->scanCG[ty][l][gi] / g;
        hm355_gen_scan(4, 4, n, ty, gx * 4, gy * 4, t->scan[ty][l] + gi * 16, 16);
      }
    }
}

static const uint8_t HM355_CHROMA_SCALE_420[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };

// what TEncSlice::setUpLambda (TEncSlice.cpp:132-159), QpParam (TComTrQuant.cpp:71-119),
// setErrScaleCoeff (:2933-2956) and the sign-hiding factor (:2382-2386) derive from (qp, lambda, weight)
static inline void hm355_fill_slice_params(FrameBuf *f, int bitDepth, int qp, double lambda, double chromaWeight)
{
  static const int quantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};
  static const int invQuantScales[6] = {40, 45, 51, 57, 64, 72};
  f->qp = qp; f->lambda = lambda; f->sqrtLambda = sqrt(lambda); f->chromaWeight = chromaWeight; f->lambdaC = lambda / chromaWeight;
  const int bdOff = 6 * (bitDepth - 8);
  const int q = qp + bdOff; f->qpPer[0] = q / 6; f->qpRem[0] = q % 6;
  int qc = qp < -bdOff ? -bdOff : (qp > 57 ? 57 : qp);
  qc = (qc < 0) ? qc + bdOff : HM355_CHROMA_SCALE_420[qc] + bdOff;
  f->qpPer[1] = qc / 6; f->qpRem[1] = qc % 6;
  for (int ch = 0; ch < 2; ch++) {
    for (int l = 0; l < 4; l++) {
      const int transformShift = 15 - bitDepth - (l + 2);
      double errScale = (double)(1 << 15);
      errScale = errScale * pow(2.0, -2.0 * transformShift);
      errScale = errScale / quantScales[f->qpRem[ch]] / quantScales[f->qpRem[ch]] / (double)(1 << (2 * (bitDepth - 8)));
      f->errScale[ch][l] = errScale;
    }
    const double lam = ch ? f->lambdaC : f->lambda;
    const double invQ = (double)invQuantScales[f->qpRem[ch]];
    f->rdFactor[ch] = (int64_t)(invQ * invQ * (double)(1 << (2 * f->qpPer[ch])) / lam / 16 / (double)(1 << (2 * (bitDepth - 8))) + 0.5);
  }
}

// I-slice lambda of an all-intra GOP (TEncSlice::initEncSlice, TEncSlice.cpp:323-352) -- used by our
// TEncSlice look-alike and by the tests; the C ABI itself takes lambda from the caller.
static inline void hm355_intra_lambda(int qp, double *lambda, double *chromaWeight)
{
  *lambda = 0.57 * pow(2.0, ((double)qp - 12) / 3.0);
  const int q = qp < 0 ? 0 : (qp > 57 ? 57 : qp);
  const int qpc = HM355_CHROMA_SCALE_420[q];
  *chromaWeight = pow(2.0, (qp - qpc) / 3.0);
}

// Dependency-ordered launch schedule.  Step s holds every CTU whose inputs (left, above-left, above,
// above-right CTU and the CABAC hand-off) were produced in steps < s:
//   WaveFrontSynchro=1 : CTU (x,y) runs at step x + 2y   (2-CTU lag wavefront, TEncSlice.cpp:740-755)
//   WaveFrontSynchro=0 : the CABAC state chains through every CTU in raster order -> step = address
// All pictures of a batch are independent, so step s carries the CTUs of every picture.
// carryLastRow (inter slices under WPP whose last CTU row is cut by the picture edge): CTU (0, last row) also needs the
// 2Nx2N integer-MV state of the last CTU of the row above, so that row is ordered behind the whole row above it.
static inline int hm355_schedule_step(int wCtu, int hCtu, int wpp, int carryLastRow, int x, int y)
{
  if (!wpp) return y * wCtu + x;
  if (carryLastRow && hCtu > 1 && wCtu > 1 && y == hCtu - 1) return (wCtu - 1) + 2 * (hCtu - 2) + 1 + x;
  return x + 2 * y;
}
static inline void hm355_build_schedule(int wCtu, int hCtu, int wpp, int nFrames, std::vector<WorkItem> &items, std::vector<int> &stepStart, int carryLastRow = 0,
                                        int firstFrame = 0, int firstRow = 0, int lastRow = -1)
{ // rows [firstRow, lastRow] of the pictures in slots [firstFrame, firstFrame + nFrames): a band of CTU rows (the rows above it are complete)
  items.clear(); stepStart.clear();
  if (lastRow < 0) lastRow = hCtu - 1;
  const int steps = hm355_schedule_step(wCtu, hCtu, wpp, carryLastRow, wCtu - 1, hCtu - 1) + 1;
  std::vector<std::vector<WorkItem> > bucket(steps);
  for (int y = firstRow; y <= lastRow; y++) for (int x = 0; x < wCtu; x++) { WorkItem w = {0, x, y, 0}; bucket[hm355_schedule_step(wCtu, hCtu, wpp, carryLastRow, x, y)].push_back(w); }
  for (int s = 0; s < steps; s++) {
    stepStart.push_back((int)items.size());
    for (int f = 0; f < nFrames; f++) for (size_t i = 0; i < bucket[s].size(); i++) { WorkItem w = bucket[s][i]; w.frame = firstFrame + f; items.push_back(w); }
  }
  stepStart.push_back((int)items.size());
}
