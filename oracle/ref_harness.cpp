// TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Harness that drives the REAL reference encoder (liron88/HM-16.2, compiled from
// /root/reference by oracle/Makefile.ref) and dumps what the hot path leaves behind,
// so that the CPU restatement in oracle/ and the HIP path can be pinned against it.
//
//   hm_dump enc  <HM options ...> --  <dump.bin>      per-CTU decision dump of every picture
//   hm_dump kat  <kat.bin> <seed>                     known-answer vectors for the primitives
//
// Reference interfaces used (nothing is copied; we only call them):
//   TAppEncTop::xInitLibCfg/xCreateLib/xInitLib   source/App/TAppEncoder/TAppEncTop.cpp:69,365,386
//   TEncTop::encode                               source/Lib/TLibEncoder/TEncTop.cpp:259
//   TComDataCU getters                            source/Lib/TLibCommon/TComDataCU.h
//   xTrMxN / xITrMxN                              source/Lib/TLibCommon/TComTrQuant.cpp:836,894
//   TComRdCost::getDistPart / setDistParam        source/Lib/TLibCommon/TComRdCost.cpp:433,356
//
// With HM_TRACE=<file> in the environment, every TComRdCost::calcRdCost call made by the
// encoder is logged (bits, distortion, cost) through a link-time --wrap, which gives an
// ordered trace of all RD decisions without touching the reference sources.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <list>
#include <vector>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <algorithm>
#include <limits>
#include <map>
#include <assert.h>
#include <stdint.h>

#define private public
#define protected public
#include "TAppEncTop.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComPic.h"
#include "TLibEncoder/TEncSlice.h"
#include "TLibEncoder/TEncCu.h"
#include "TLibEncoder/TEncPic.h"
#undef private
#undef protected

// external-linkage functions of TComTrQuant.cpp (not declared in a header)
Void xTrMxN(Int bitDepth, TCoeff *block, TCoeff *coeff, Int iWidth, Int iHeight, Bool useDST, const Int maxTrDynamicRange);
Void xITrMxN(Int bitDepth, TCoeff *coeff, TCoeff *block, Int iWidth, Int iHeight, Bool useDST, const Int maxTrDynamicRange);

// ---------------------------------------------------------------------------------------------
// link-time trace of calcRdCost
// ---------------------------------------------------------------------------------------------
static FILE *g_trace = NULL;
extern "C" Double __real__ZN10TComRdCost10calcRdCostEjjb5DFunc(TComRdCost *self, UInt bits, UInt dist, Bool flag, DFunc f);
extern "C" Double __wrap__ZN10TComRdCost10calcRdCostEjjb5DFunc(TComRdCost *self, UInt bits, UInt dist, Bool flag, DFunc f)
{
  Double c = __real__ZN10TComRdCost10calcRdCostEjjb5DFunc(self, bits, dist, flag, f);
  if (g_trace) fprintf(g_trace, "RD %u %u %.1f\n", bits, dist, c);
  return c;
}

// ---------------------------------------------------------------------------------------------
// dump format (little endian):
//   header : "HMD1", u32 width, height, bitDepth, ctuSize, numFrames
//   frame  : u32 poc, u32 numCtus, then per CTU (raster order):
//              f64 totalCost, u32 totalBits, u32 totalDist,
//              u8[256] depth, partSize, predMode, intraDirLuma, intraDirChroma, trIdx,
//              cbfY, cbfCb, cbfCr, tskipY, tskipCb, tskipCr     (z-scan order, 4x4 units)
//              i32[4096] coeffY, i32[1024] coeffCb, i32[1024] coeffCr  (HM's own TU packing)
//            then the reconstructed planes of the picture (u16, w*h, w/2*h/2 *2), no margins
// ---------------------------------------------------------------------------------------------
static void put32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }

static void dumpPic(FILE *f, TComPic *pic)
{
  TComPicSym *sym = pic->getPicSym();
  const UInt numCtus = sym->getNumberOfCtusInFrame();
  put32(f, (uint32_t)pic->getPOC());
  put32(f, numCtus);
  for (UInt a = 0; a < numCtus; a++)
  {
    TComDataCU *ctu = pic->getCtu(a);
    double cost = ctu->getTotalCost();
    fwrite(&cost, 8, 1, f);
    put32(f, ctu->getTotalBits());
    put32(f, (uint32_t)ctu->getTotalDistortion());
    const UInt np = pic->getNumPartitionsInCtu();
    assert(np == 256);
    uint8_t buf[12][256];
    for (UInt i = 0; i < np; i++)
    {
      buf[0][i] = ctu->getDepth(i);
      buf[1][i] = (uint8_t)ctu->getPartitionSize(i);
      buf[2][i] = (uint8_t)ctu->getPredictionMode(i);
      buf[3][i] = ctu->getIntraDir(CHANNEL_TYPE_LUMA, i);
      buf[4][i] = ctu->getIntraDir(CHANNEL_TYPE_CHROMA, i);
      buf[5][i] = ctu->getTransformIdx(i);
      buf[6][i] = ctu->getCbf(COMPONENT_Y)[i];
      buf[7][i] = ctu->getCbf(COMPONENT_Cb)[i];
      buf[8][i] = ctu->getCbf(COMPONENT_Cr)[i];
      buf[9][i] = ctu->getTransformSkip(COMPONENT_Y)[i];
      buf[10][i] = ctu->getTransformSkip(COMPONENT_Cb)[i];
      buf[11][i] = ctu->getTransformSkip(COMPONENT_Cr)[i];
    }
    fwrite(buf, 1, sizeof(buf), f);
    fwrite(ctu->getCoeff(COMPONENT_Y), 4, 4096, f);
    fwrite(ctu->getCoeff(COMPONENT_Cb), 4, 1024, f);
    fwrite(ctu->getCoeff(COMPONENT_Cr), 4, 1024, f);
  }
  TComPicYuv *rec = pic->getPicYuvRec();
  for (int c = 0; c < 3; c++)
  {
    ComponentID id = ComponentID(c);
    const Int w = rec->getWidth(id), h = rec->getHeight(id), s = rec->getStride(id);
    const Pel *p = rec->getAddr(id);
    std::vector<uint16_t> row(w);
    for (Int y = 0; y < h; y++)
    {
      for (Int x = 0; x < w; x++) row[x] = (uint16_t)p[y * s + x];
      fwrite(&row[0], 2, w, f);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// inter dump (HMD2): link-time wraps around TEncSlice::compressSlice (TEncSlice.cpp:640) and
// TComPic::compressMotion (TComPic.cpp, called at the end of each picture, TEncGOP.cpp:1660).
// Record stream (little endian), in encoding order:
//   'S' : one slice as compressSlice sees and leaves it
//         i32 poc, sliceType, sliceQp, tLayer, depth; f64 lambda, sqrtLambda, weightCb, weightCr;
//         u32 lambdaMotionSAD, lambdaMotionSSE; i32 numRefIdx[2]; i32 refPOC[2][16]; i32 refIsLongTerm[2][16];
//         i32 colFromL0, colRefIdx, enableTMVP, mvdL1Zero, maxNumMergeCand, checkLDC, cabacInitType; i32 list1ToList0[16]
//         u32 numCtus; per CTU: f64 cost, u32 bits, u32 dist, u8[12][256] (as HMD1), u8[4][256] skip, mergeFlag,
//         mergeIndex, interDir; per list l: i16[256][2] mv, i16[256][2] mvd, i8[256] refIdx, i8[256] mvpIdx, i8[256] mvpNum;
//         i32[6144] coefficients; then the pre-deblocking reconstruction planes (u16)
//   'F' : the finished picture as later pictures reference it
//         i32 poc; final reconstruction planes (u16, after deblocking + SAO); i32 sliceType; i32 numRefIdx[2];
//         i32 refPOC[2][16]; i32 refIsLongTerm[2][16]; u32 numCtus; per CTU: u8[256] predMode and, per list, i16[256][2] mv,
//         i8[256] refIdx -- the motion field after compressMotion (16x16 granularity)
// ---------------------------------------------------------------------------------------------
static FILE *g_dump2 = NULL;
static void putPlanes(FILE *f, TComPicYuv *rec)
{
  for (int c = 0; c < 3; c++)
  {
    ComponentID id = ComponentID(c);
    const Int w = rec->getWidth(id), h = rec->getHeight(id), s = rec->getStride(id);
    const Pel *p = rec->getAddr(id);
    std::vector<uint16_t> row(w);
    for (Int y = 0; y < h; y++) { for (Int x = 0; x < w; x++) row[x] = (uint16_t)p[y * s + x]; fwrite(&row[0], 2, w, f); }
  }
}
static void putRefLists(FILE *f, TComSlice *sl)
{
  int32_t n[2] = { sl->getSliceType() == I_SLICE ? 0 : sl->getNumRefIdx(REF_PIC_LIST_0), sl->getSliceType() == B_SLICE ? sl->getNumRefIdx(REF_PIC_LIST_1) : 0 };
  fwrite(n, 4, 2, f);
  int32_t poc[2][16], lt[2][16];
  memset(poc, 0, sizeof(poc)); memset(lt, 0, sizeof(lt));
  for (int l = 0; l < 2; l++) for (int i = 0; i < n[l]; i++)
  { poc[l][i] = sl->getRefPOC(RefPicList(l), i); lt[l][i] = sl->getRefPic(RefPicList(l), i)->getIsLongTerm() ? 1 : 0; }
  fwrite(poc, 4, 32, f); fwrite(lt, 4, 32, f);
}
static void putMotion(FILE *f, TComDataCU *ctu, int l, bool full)
{
  TComCUMvField *mf = ctu->getCUMvField(RefPicList(l));
  int16_t mv[256][2], mvd[256][2]; int8_t ri[256], mi[256], mn[256];
  for (int i = 0; i < 256; i++)
  {
    mv[i][0] = (int16_t)mf->getMv(i).getHor(); mv[i][1] = (int16_t)mf->getMv(i).getVer();
    mvd[i][0] = (int16_t)mf->getMvd(i).getHor(); mvd[i][1] = (int16_t)mf->getMvd(i).getVer();
    ri[i] = (int8_t)mf->getRefIdx(i); mi[i] = (int8_t)ctu->getMVPIdx(RefPicList(l), i); mn[i] = (int8_t)ctu->getMVPNum(RefPicList(l), i);
  }
  fwrite(mv, 2, 512, f);
  if (full) fwrite(mvd, 2, 512, f);
  fwrite(ri, 1, 256, f);
  if (full) { fwrite(mi, 1, 256, f); fwrite(mn, 1, 256, f); }
}
// TEncCu::compressCtu: the lambda every CTU is searched with (the LCU-level rate control sets one per CTU, TEncSlice.cpp:776-808)
static std::vector<double> g_ctuLambda;
static std::vector<int32_t> g_ctuRcQp;
extern "C" void __real__ZN6TEncCu11compressCtuEP10TComDataCU(TEncCu *self, TComDataCU *ctu);
extern "C" void __wrap__ZN6TEncCu11compressCtuEP10TComDataCU(TEncCu *self, TComDataCU *ctu)
{
  g_ctuLambda.push_back(self->m_pcRdCost->m_dLambda);
  g_ctuRcQp.push_back(self->m_pcEncCfg->getUseRateCtrl() ? (int32_t)self->m_pcRateCtrl->getRCQP() : 0);      // what xCompressCU takes as the CTU's QP (TEncCu.cpp:522-526)
  __real__ZN6TEncCu11compressCtuEP10TComDataCU(self, ctu);
}
extern "C" void __real__ZN9TEncSlice13compressSliceEP7TComPic(TEncSlice *self, TComPic *pic);
extern "C" void __wrap__ZN9TEncSlice13compressSliceEP7TComPic(TEncSlice *self, TComPic *pic)
{
  const int32_t dqpFlagIn = self->m_pcCuEncoder->getdQPFlag() ? 1 : 0;      // TEncCu::m_bEncodeDQP as the previous picture's encodeSlice left it
  g_ctuLambda.clear(); g_ctuRcQp.clear();
  __real__ZN9TEncSlice13compressSliceEP7TComPic(self, pic);
  if (!g_dump2) return;
  FILE *f = g_dump2;
  TComSlice *sl = pic->getSlice(self->getSliceIdx());
  if (self->m_pcCfg->getUseRateCtrl() && self->m_pcCfg->getLCULevelRC())
  { // 'L' (before the 'Q' / 'S' records of the same slice, only under the LCU-level rate control): u32 numCtus; f64 lambda of every CTU's search; i32 its QP
    fputc('L', f);
    put32(f, (uint32_t)g_ctuLambda.size());
    fwrite(g_ctuLambda.data(), 8, g_ctuLambda.size(), f);
    fwrite(g_ctuRcQp.data(), 4, g_ctuRcQp.size(), f);
  }
  if (sl->getPPS()->getUseDQP())
  { // 'Q' (before the 'S' record of the same slice, only when cu_qp_delta is enabled): i32 maxCuDQPDepth, dqpFlagIn, dqpFlagOut, qpAdaptationRange;
    //   u32 numCtus; per CTU i8[256] m_phQP as compressSlice left it; then the layer-0 activities of TEncPreanalyzer when AdaptiveQP is on:
    //   u32 numUnits (0: off), f64 avgActivity, f64 activity[numUnits]
    fputc('Q', f);
    int32_t q[4] = { (int32_t)sl->getPPS()->getMaxCuDQPDepth(), dqpFlagIn, self->m_pcCuEncoder->getdQPFlag() ? 1 : 0, self->m_pcCfg->getUseAdaptiveQP() ? self->m_pcCfg->getQPAdaptationRange() : 0 };
    fwrite(q, 4, 4, f);
    const UInt nc = pic->getPicSym()->getNumberOfCtusInFrame();
    put32(f, nc);
    for (UInt a = 0; a < nc; a++) { int8_t qp[256]; for (int i = 0; i < 256; i++) qp[i] = (int8_t)pic->getCtu(a)->getQP(i); fwrite(qp, 1, 256, f); }
    TEncPic *ep = dynamic_cast<TEncPic *>(pic);
    if (self->m_pcCfg->getUseAdaptiveQP() && ep && ep->getMaxAQDepth() > 0)
    {
      TEncPicQPAdaptationLayer *lay = ep->getAQLayer(0);
      const UInt nu = lay->getNumAQPartInWidth() * lay->getNumAQPartInHeight();
      put32(f, nu);
      double avg = lay->getAvgActivity(); fwrite(&avg, 8, 1, f);
      for (UInt i = 0; i < nu; i++) { double act = lay->getQPAdaptationUnit()[i].getActivity(); fwrite(&act, 8, 1, f); }
    }
    else put32(f, 0);
  }
  fputc('S', f);
  int32_t hdr[5] = { sl->getPOC(), (int32_t)sl->getSliceType(), sl->getSliceQp(), (int32_t)sl->getTLayer(), sl->getDepth() };
  fwrite(hdr, 4, 5, f);
  double d[4] = { self->m_pcRdCost->m_dLambda, self->m_pcRdCost->m_sqrtLambda, self->m_pcRdCost->m_distortionWeight[COMPONENT_Cb], self->m_pcRdCost->m_distortionWeight[COMPONENT_Cr] };
  fwrite(d, 8, 4, f);
  put32(f, self->m_pcRdCost->m_uiLambdaMotionSAD[0]); put32(f, self->m_pcRdCost->m_uiLambdaMotionSSE[0]);
  putRefLists(f, sl);
  // context initialisation table the estimator used (TEncSbac::resetEntropy, TEncSbac.cpp:106-115)
  int32_t initType = (int32_t)sl->getSliceType();
  { const Int idx = sl->getPPS()->getEncCABACTableIdx();
    if (!sl->isIntra() && (idx == B_SLICE || idx == P_SLICE) && sl->getPPS()->getCabacInitPresentFlag()) initType = idx; }
  int32_t misc[7] = { (int32_t)sl->getColFromL0Flag(), (int32_t)sl->getColRefIdx(), sl->getEnableTMVPFlag() ? 1 : 0, sl->getMvdL1ZeroFlag() ? 1 : 0,
                      (int32_t)sl->getMaxNumMergeCand(), sl->getCheckLDC() ? 1 : 0, initType };
  fwrite(misc, 4, 7, f);
  int32_t l1l0[16]; for (int i = 0; i < 16; i++) l1l0[i] = sl->getList1IdxToList0Idx(i);
  fwrite(l1l0, 4, 16, f);
  const UInt numCtus = pic->getPicSym()->getNumberOfCtusInFrame();
  put32(f, numCtus);
  for (UInt a = 0; a < numCtus; a++)
  {
    TComDataCU *ctu = pic->getCtu(a);
    double cost = ctu->getTotalCost();
    fwrite(&cost, 8, 1, f); put32(f, ctu->getTotalBits()); put32(f, (uint32_t)ctu->getTotalDistortion());
    uint8_t buf[16][256];
    for (UInt i = 0; i < 256; i++)
    {
      buf[0][i] = ctu->getDepth(i); buf[1][i] = (uint8_t)ctu->getPartitionSize(i); buf[2][i] = (uint8_t)ctu->getPredictionMode(i);
      buf[3][i] = ctu->getIntraDir(CHANNEL_TYPE_LUMA, i); buf[4][i] = ctu->getIntraDir(CHANNEL_TYPE_CHROMA, i); buf[5][i] = ctu->getTransformIdx(i);
      buf[6][i] = ctu->getCbf(COMPONENT_Y)[i]; buf[7][i] = ctu->getCbf(COMPONENT_Cb)[i]; buf[8][i] = ctu->getCbf(COMPONENT_Cr)[i];
      buf[9][i] = ctu->getTransformSkip(COMPONENT_Y)[i]; buf[10][i] = ctu->getTransformSkip(COMPONENT_Cb)[i]; buf[11][i] = ctu->getTransformSkip(COMPONENT_Cr)[i];
      buf[12][i] = ctu->getSkipFlag(i) ? 1 : 0; buf[13][i] = ctu->getMergeFlag(i) ? 1 : 0; buf[14][i] = ctu->getMergeIndex(i); buf[15][i] = ctu->getInterDir(i);
    }
    fwrite(buf, 1, sizeof(buf), f);
    putMotion(f, ctu, 0, true); putMotion(f, ctu, 1, true);
    fwrite(ctu->getCoeff(COMPONENT_Y), 4, 4096, f); fwrite(ctu->getCoeff(COMPONENT_Cb), 4, 1024, f); fwrite(ctu->getCoeff(COMPONENT_Cr), 4, 1024, f);
  }
  putPlanes(f, pic->getPicYuvRec());
}
// 'B' : what TEncSlice::encodeSlice (TEncSlice.cpp:910) wrote for the slice, before the substreams are concatenated into the NAL unit
//       (TEncGOP.cpp:1559-1583): i32 poc, u32 numSubstreams, per substream u32 numBytes + the bytes, then i32 encCABACTableIdx as
//       determineCabacInitIdx left it for the following pictures (TEncSbac.cpp:163) and u32 numBinsCoded
extern "C" void __real__ZN9TEncSlice11encodeSliceEP7TComPicP19TComOutputBitstreamRj(TEncSlice *self, TComPic *pic, TComOutputBitstream *subs, UInt &numBins);
extern "C" void __wrap__ZN9TEncSlice11encodeSliceEP7TComPicP19TComOutputBitstreamRj(TEncSlice *self, TComPic *pic, TComOutputBitstream *subs, UInt &numBins)
{
  __real__ZN9TEncSlice11encodeSliceEP7TComPicP19TComOutputBitstreamRj(self, pic, subs, numBins);
  if (!g_dump2) return;
  FILE *f = g_dump2;
  TComSlice *sl = pic->getSlice(self->getSliceIdx());
  fputc('B', f);
  int32_t poc = sl->getPOC(); fwrite(&poc, 4, 1, f);
  const UInt n = (UInt)sl->getPPS()->getNumSubstreams();
  put32(f, n);
  for (UInt i = 0; i < n; i++)
  {
    const std::vector<uint8_t> &b = subs[i].getFIFO();
    put32(f, (uint32_t)b.size());
    if (!b.empty()) fwrite(&b[0], 1, b.size(), f);
  }
  int32_t idx = (int32_t)sl->getPPS()->getEncCABACTableIdx(); fwrite(&idx, 4, 1, f);
  put32(f, numBins);
}
extern "C" void __real__ZN7TComPic14compressMotionEv(TComPic *self);
extern "C" void __wrap__ZN7TComPic14compressMotionEv(TComPic *self)
{
  __real__ZN7TComPic14compressMotionEv(self);
  if (!g_dump2) return;
  FILE *f = g_dump2;
  TComSlice *sl = self->getSlice(0);
  fputc('F', f);
  int32_t poc = self->getPOC(); fwrite(&poc, 4, 1, f);
  putPlanes(f, self->getPicYuvRec());
  int32_t st = (int32_t)sl->getSliceType(); fwrite(&st, 4, 1, f);
  putRefLists(f, sl);
  const UInt numCtus = self->getPicSym()->getNumberOfCtusInFrame();
  put32(f, numCtus);
  for (UInt a = 0; a < numCtus; a++)
  {
    TComDataCU *ctu = self->getCtu(a);
    uint8_t pm[256]; for (int i = 0; i < 256; i++) pm[i] = (uint8_t)ctu->getPredictionMode(i);
    fwrite(pm, 1, 256, f);
    putMotion(f, ctu, 0, false); putMotion(f, ctu, 1, false);
  }
  // 'A' : what TEncSampleAdaptiveOffset::SAOProcess decided for this picture (TEncGOP.cpp:1483): i32 poc, i32 depth (temporal depth the
  //       picture-level on/off rule uses), i32 enabled[2] (luma, chroma slice flags), u32 numCtus, then per CTU and component
  //       i32 modeIdc, typeIdc, typeAuxInfo, offset[32]  (SAOBlkParam of TComPicSym)
  fputc('A', f);
  fwrite(&poc, 4, 1, f);
  int32_t dep = (int32_t)sl->getDepth(); fwrite(&dep, 4, 1, f);
  int32_t en[2] = { (int32_t)sl->getSaoEnabledFlag(CHANNEL_TYPE_LUMA), (int32_t)sl->getSaoEnabledFlag(CHANNEL_TYPE_CHROMA) }; fwrite(en, 4, 2, f);
  put32(f, numCtus);
  SAOBlkParam *sao = self->getPicSym()->getSAOBlkParam();
  for (UInt a = 0; a < numCtus; a++)
    for (int c = 0; c < 3; c++)
    {
      const SAOOffset &o = sao[a][c];
      int32_t v[35]; v[0] = o.modeIdc; v[1] = o.typeIdc; v[2] = o.typeAuxInfo; for (int k = 0; k < 32; k++) v[3 + k] = o.offset[k];
      fwrite(v, 4, 35, f);
    }
}

static int runEnc(int argc, char **argv, const char *dumpName)
{
  TAppEncTop app;
  app.create();
  if (!app.parseCfg(argc, argv)) { app.destroy(); return 1; }

  FILE *df = fopen(dumpName, "wb");
  if (!df) { perror(dumpName); return 1; }
  fwrite("HMD1", 1, 4, df);
  put32(df, app.m_iSourceWidth); put32(df, app.m_iSourceHeight);
  put32(df, app.m_internalBitDepth[0]); put32(df, app.m_uiMaxCUWidth); put32(df, app.m_framesToBeEncoded);

  std::fstream bitstreamFile(app.m_pchBitstreamFile, std::fstream::binary | std::fstream::out);
  TComPicYuv *pcPicYuvOrg = new TComPicYuv;
  TComPicYuv *pcPicYuvRec = NULL;
  app.xInitLibCfg();
  app.xCreateLib();
  app.xInitLib(app.m_isField);

  Int iNumEncoded = 0;
  Bool bEos = false;
  const InputColourSpaceConversion ipCSC = app.m_inputColourSpaceConvert;
  const InputColourSpaceConversion snrCSC = (!app.m_snrInternalColourSpace) ? app.m_inputColourSpaceConvert : IPCOLOURSPACE_UNCHANGED;
  std::list<AccessUnit> outputAccessUnits;
  TComPicYuv cPicYuvTrueOrg;
  pcPicYuvOrg->create(app.m_iSourceWidth, app.m_iSourceHeight, app.m_chromaFormatIDC, app.m_uiMaxCUWidth, app.m_uiMaxCUHeight, app.m_uiMaxCUDepth);
  cPicYuvTrueOrg.create(app.m_iSourceWidth, app.m_iSourceHeight, app.m_chromaFormatIDC, app.m_uiMaxCUWidth, app.m_uiMaxCUHeight, app.m_uiMaxCUDepth);

  int nextPocToDump = 0;
  while (!bEos)
  {
    app.xGetBuffer(pcPicYuvRec);
    app.m_cTVideoIOYuvInputFile.read(pcPicYuvOrg, &cPicYuvTrueOrg, ipCSC, app.m_aiPad, app.m_InputChromaFormatIDC);
    app.m_iFrameRcvd++;
    bEos = (app.m_iFrameRcvd == app.m_framesToBeEncoded);
    Bool flush = 0;
    if (app.m_cTVideoIOYuvInputFile.isEof())
    {
      flush = true; bEos = true; app.m_iFrameRcvd--;
      app.m_cTEncTop.setFramesToBeEncoded(app.m_iFrameRcvd);
    }
    app.m_cTEncTop.encode(bEos, flush ? 0 : pcPicYuvOrg, flush ? 0 : &cPicYuvTrueOrg, snrCSC, app.m_cListPicYuvRec, outputAccessUnits, iNumEncoded);
    if (iNumEncoded > 0)
    {
      // every picture just encoded is still in the encoder's picture list, with its per-CTU data
      TComList<TComPic*> *lst = app.m_cTEncTop.getListPic();
      for (int k = 0; k < iNumEncoded; k++)
      {
        for (TComList<TComPic*>::iterator it = lst->begin(); it != lst->end(); ++it)
        {
          if ((*it)->getPOC() == nextPocToDump && (*it)->getReconMark()) { dumpPic(df, *it); break; }
        }
        nextPocToDump++;
      }
      app.xWriteOutput(bitstreamFile, iNumEncoded, outputAccessUnits);
      outputAccessUnits.clear();
    }
  }
  app.m_cTEncTop.printSummary(app.m_isField);
  fclose(df);
  // leak the rest on purpose: process exits
  return 0;
}

// ---------------------------------------------------------------------------------------------
// primitive known-answer vectors
//   "KAT1", then records: u32 tag, u32 n_in, i32[n_in] params+inputs, u32 n_out, i32[n_out]
//   tags: 1 SAD, 2 SSE, 3 HAD, 4 fwd transform, 5 inv transform
// ---------------------------------------------------------------------------------------------
static uint32_t g_rng = 1;
static uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5; return g_rng; }

static void putRec(FILE *f, uint32_t tag, const std::vector<int32_t> &in, const std::vector<int32_t> &out)
{
  put32(f, tag); put32(f, (uint32_t)in.size()); fwrite(in.data(), 4, in.size(), f);
  put32(f, (uint32_t)out.size()); fwrite(out.data(), 4, out.size(), f);
}

static int runKat(const char *name, uint32_t seed)
{
  g_rng = seed ? seed : 1;
  FILE *f = fopen(name, "wb");
  if (!f) { perror(name); return 1; }
  fwrite("KAT1", 1, 4, f);
  initROM();
  TComRdCost rd; rd.init();
  const int sizes[5] = {4, 8, 16, 32, 64};
  for (int bd = 8; bd <= 10; bd += 2)
  {
    g_bitDepth[0] = g_bitDepth[1] = bd;
    g_maxTrDynamicRange[0] = g_maxTrDynamicRange[1] = 15;
    const int maxv = (1 << bd) - 1;
    for (int si = 0; si < 5; si++)
    {
      const int n = sizes[si];
      for (int rep = 0; rep < 6; rep++)
      {
        std::vector<Pel> a(n * n), b(n * n);
        const int spread = (rep < 2) ? maxv : (rep < 4 ? 40 : 6);
        for (int i = 0; i < n * n; i++)
        {
          int base = rnd() % (maxv + 1);
          a[i] = (Pel)base;
          int d = (int)(rnd() % (2 * spread + 1)) - spread;
          b[i] = (Pel)std::min(maxv, std::max(0, base + d));
        }
        std::vector<int32_t> in;
        in.push_back(bd); in.push_back(n);
        for (int i = 0; i < n * n; i++) in.push_back(a[i]);
        for (int i = 0; i < n * n; i++) in.push_back(b[i]);
        // SAD with sub-shift 0 and 1 (the FEN row subsampling of integer ME)
        for (int sub = 0; sub < 2; sub++)
        {
          DistParam dp;
          dp.pOrg = &a[0]; dp.pCur = &b[0]; dp.iStrideOrg = n; dp.iStrideCur = n; dp.iCols = n; dp.iRows = n;
          dp.iStep = 1; dp.bApplyWeight = false; dp.bitDepth = bd; dp.iSubShift = sub;
          rd.setDistParam(n, n, DF_SAD, dp);
          dp.iSubShift = sub;
          std::vector<int32_t> in2(in); in2.insert(in2.begin() + 2, sub);
          std::vector<int32_t> out(1, (int32_t)dp.DistFunc(&dp));
          putRec(f, 1, in2, out);
        }
        {
          std::vector<int32_t> out(1, (int32_t)rd.getDistPart(bd, &b[0], n, &a[0], n, n, n, COMPONENT_Y, DF_SSE));
          putRec(f, 2, in, out);
        }
        {
          std::vector<int32_t> out(1, (int32_t)rd.getDistPart(bd, &b[0], n, &a[0], n, n, n, COMPONENT_Y, DF_HADS));
          putRec(f, 3, in, out);
        }
        if (n <= 32)
        {
          for (int dst = 0; dst < (n == 4 ? 2 : 1); dst++)
          {
            std::vector<TCoeff> blk(n * n), coef(n * n), back(n * n);
            for (int i = 0; i < n * n; i++) blk[i] = (TCoeff)a[i] - (TCoeff)b[i];
            xTrMxN(bd, &blk[0], &coef[0], n, n, dst != 0, 15);
            std::vector<int32_t> tin; tin.push_back(bd); tin.push_back(n); tin.push_back(dst);
            for (int i = 0; i < n * n; i++) tin.push_back(blk[i]);
            // xTrMxN clobbers nothing in blk; record output
            std::vector<int32_t> tout(coef.begin(), coef.end());
            putRec(f, 4, tin, tout);
            // inverse of a sparsified / quantised-looking version
            std::vector<TCoeff> q(n * n);
            for (int i = 0; i < n * n; i++) q[i] = (rnd() % 4 == 0) ? (coef[i] / 8) * 8 : 0;
            std::vector<TCoeff> qc(q);
            xITrMxN(bd, &qc[0], &back[0], n, n, dst != 0, 15);
            std::vector<int32_t> iin; iin.push_back(bd); iin.push_back(n); iin.push_back(dst);
            for (int i = 0; i < n * n; i++) iin.push_back(q[i]);
            std::vector<int32_t> iout(back.begin(), back.end());
            putRec(f, 5, iin, iout);
          }
        }
      }
    }
  }
  fclose(f);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// yuvio: the reference's own file reader / writer (TVideoIOYuv::read :633, ::write :706) as a known-answer generator for the ingest row
// (SURVEY 8f n3).  hm_dump yuvio <in.yuv> <fileW> <fileH> <fileBitDepth> <internalBitDepth> <padX> <padY> <frames> <outBitDepth> <out.yuv> <dump.bin>
// reads every frame into a (fileW + padX) x (fileH + padY) 4:2:0 picture exactly as TAppEncTop::encode does (:407-455), dumps its planes
// (u16, tightly packed) and writes it back through TVideoIOYuv::write with the padding as conformance window (:603) into out.yuv.
// ---------------------------------------------------------------------------------------------
static int runYuvIo(char **a)
{
  const int fw = atoi(a[1]), fh = atoi(a[2]), fbd = atoi(a[3]), ibd = atoi(a[4]), px = atoi(a[5]), py = atoi(a[6]), nf = atoi(a[7]), obd = atoi(a[8]);
  Int fileBD[2] = { fbd, fbd }, intBD[2] = { ibd, ibd }, outBD[2] = { obd, obd }, pad[2] = { px, py };
  TVideoIOYuv in, out;
  in.open(a[0], false, fileBD, fileBD, intBD);
  out.open(a[9], true, outBD, outBD, intBD);
  TComPicYuv org, trueOrg;
  org.create(fw + px, fh + py, CHROMA_420, 64, 64, 4); trueOrg.create(fw + px, fh + py, CHROMA_420, 64, 64, 4);
  FILE *f = fopen(a[10], "wb");
  if (!f) { perror(a[10]); return 1; }
  for (int i = 0; i < nf; i++)
  {
    if (!in.read(&org, &trueOrg, IPCOLOURSPACE_UNCHANGED, pad, CHROMA_420)) { fprintf(stderr, "yuvio: short read\n"); return 1; }
    putPlanes(f, &org);
    if (!out.write(&org, IPCOLOURSPACE_UNCHANGED, 0, px, 0, py)) { fprintf(stderr, "yuvio: write failed\n"); return 1; }
  }
  fclose(f); in.close(); out.close();
  org.destroy(); trueOrg.destroy();
  return 0;
}

int main(int argc, char **argv)
{
  if (argc >= 13 && !strcmp(argv[1], "yuvio")) return runYuvIo(argv + 2);
  if (getenv("HM_TRACE")) g_trace = fopen(getenv("HM_TRACE"), "w");
  if (argc >= 4 && !strcmp(argv[1], "kat")) return runKat(argv[2], (uint32_t)atoi(argv[3]));
  if (argc >= 4 && !strcmp(argv[1], "enc2"))
  { // hm_dump enc2 <HM options...> -- dump2.bin : the HMD2 record stream (inter-capable)
    int sep = -1;
    for (int i = 2; i < argc; i++) if (!strcmp(argv[i], "--")) sep = i;
    if (sep < 0 || sep + 1 >= argc) { fprintf(stderr, "usage: hm_dump enc2 <HM options> -- <dump2.bin>\n"); return 2; }
    std::vector<char*> av; av.push_back(argv[0]);
    for (int i = 2; i < sep; i++) av.push_back(argv[i]);
    g_dump2 = fopen(argv[sep + 1], "wb");
    if (!g_dump2) { perror(argv[sep + 1]); return 1; }
    fwrite("HMD2", 1, 4, g_dump2);
    int rc = runEnc((int)av.size(), &av[0], "/dev/null");
    fclose(g_dump2);
    if (g_trace) fclose(g_trace);
    return rc;
  }
  if (argc >= 4 && !strcmp(argv[1], "enc"))
  {
    // argv: hm_dump enc <HM options...> -- dump.bin
    int sep = -1;
    for (int i = 2; i < argc; i++) if (!strcmp(argv[i], "--")) sep = i;
    if (sep < 0 || sep + 1 >= argc) { fprintf(stderr, "usage: hm_dump enc <HM options> -- <dump.bin>\n"); return 2; }
    std::vector<char*> av; av.push_back(argv[0]);
    for (int i = 2; i < sep; i++) av.push_back(argv[i]);
    int rc = runEnc((int)av.size(), &av[0], argv[sep + 1]);
    if (g_trace) fclose(g_trace);
    return rc;
  }
  fprintf(stderr, "usage: hm_dump enc <HM options> -- <dump.bin> | hm_dump kat <kat.bin> <seed>\n");
  return 2;
}
