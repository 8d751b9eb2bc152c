#include "TEncTop.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const unsigned char kChromaScale420[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };

Void TComPic::create(Int w, Int h) { m_org.create(w, h); m_rec.create(w, h); m_ctus.assign((size_t)((w + 63) / 64) * ((h + 63) / 64), hm355_ctu_out()); }

// ---- TEncTop ----
Void TEncTop::create()
{
  hm355_seq_cfg cfg; memset(&cfg, 0, sizeof(cfg));
  cfg.width = m_iSourceWidth; cfg.height = m_iSourceHeight; cfg.bit_depth = m_bitDepth;
  cfg.ctu_size = 64; cfg.max_cu_depth = 4; cfg.tu_log2_max = 5; cfg.tu_log2_min = 2; cfg.tu_max_depth_intra = 3;
  cfg.wavefront_synchro = m_iWaveFrontSynchro; cfg.max_batch = 1;
  const int rc = hm355_create(&cfg, &m_ctx);
  if (rc != HM355_OK) { // the reference reports fatal set-up errors with exit(), TAppEncCfg.cpp:1422-2195
    fprintf(stderr, "TEncTop::create: hm355_create failed (%d) %s\n", rc, m_ctx ? hm355_last_error(m_ctx) : "");
    if (m_ctx) hm355_destroy(m_ctx);
    exit(EXIT_FAILURE);
  }
}
Void TEncTop::destroy() { for (auto p : m_cListPic) delete p; m_cListPic.clear(); if (m_ctx) hm355_destroy(m_ctx); m_ctx = nullptr; }
Void TEncTop::init() { m_cGOPEncoder.init(this); m_cSliceEncoder.init(this); m_cLoopFilter.init(this); m_cEncSAO.init(this); }
Void TEncTop::encode(Bool flush, TComPicYuv *pcPicYuvOrg, std::list<TComPic *> &rcListPicOut, Int &iNumEncoded)
{
  iNumEncoded = 0;
  if (pcPicYuvOrg) {
    TComPic *pic = new TComPic; pic->create(m_iSourceWidth, m_iSourceHeight);
    for (Int c = 0; c < 3; c++)
      memcpy(pic->getPicYuvOrg()->getAddr(ComponentID(c)), pcPicYuvOrg->getAddr(ComponentID(c)),
             sizeof(uint16_t) * pcPicYuvOrg->getWidth(ComponentID(c)) * pcPicYuvOrg->getHeight(ComponentID(c)));
    m_iPOCLast++; m_iNumPicRcvd++;
    pic->getSlice(0)->setPOC(m_iPOCLast);
    m_cListPic.push_back(pic);
  }
  if (!m_iNumPicRcvd || (!flush && m_iNumPicRcvd != m_iGOPSize)) return;   // TEncTop.cpp:277
  m_cGOPEncoder.compressGOP(m_iPOCLast, m_iNumPicRcvd, m_cListPic);
  iNumEncoded = m_iNumPicRcvd; m_iNumPicRcvd = 0;
  rcListPicOut = m_cListPic;
}

// ---- TEncGOP ----
Void TEncGOP::init(TEncTop *t) { m_pcEncTop = t; m_pcSliceEncoder = t->getSliceEncoder(); m_pcLoopFilter = t->getLoopFilter(); m_pcSAO = t->getSAO(); }
Void TEncGOP::compressGOP(Int iPOCLast, Int iNumPicRcvd, std::list<TComPic *> &rcListPic)
{
  Int iGOPid = 0;
  for (auto it = rcListPic.rbegin(); it != rcListPic.rend() && iGOPid < iNumPicRcvd; ++it, ++iGOPid) {
    TComPic *pcPic = *it; TComSlice *pcSlice = nullptr;
    m_pcSliceEncoder->initEncSlice(pcPic, iPOCLast, pcPic->getPOC(), iNumPicRcvd, iGOPid, pcSlice);   // TEncGOP.cpp:760
    m_pcSliceEncoder->precompressSlice(pcPic);                                                        // :1137
    m_pcSliceEncoder->compressSlice(pcPic);                                                           // :1138
    pcSlice->setSliceBits((UInt)m_pcSliceEncoder->getTotalBits());
    // loop filters (TEncGOP.cpp:1184, :1475-1497), then the finished picture back into getPicYuvRec()
    const Bool bLF = !m_pcEncTop->getLoopFilterDisable(), bSAO = m_pcEncTop->getUseSAO();
    if (bLF) m_pcLoopFilter->loopFilterPic(pcPic);
    pcSlice->setSaoEnabledFlag(0, false); pcSlice->setSaoEnabledFlag(1, false);
    if (bSAO) {
      Bool sliceEnabled[3]; m_pcSAO->SAOProcess(pcPic, sliceEnabled, m_pcSliceEncoder->getLambdas());
      pcSlice->setSaoEnabledFlag(0, sliceEnabled[0]); pcSlice->setSaoEnabledFlag(1, sliceEnabled[1]);      // TEncGOP.cpp:1486-1490
    }
    { // the slice data (TEncGOP.cpp:1127, :1556-1561): one substream per CTU row with WPP, else one
      const Int numSubstreams = hm355_num_substreams(m_pcEncTop->getDeviceContext());
      std::vector<TComOutputBitstream> &substreamsOut = pcPic->getSubstreams();
      substreamsOut.assign(numSubstreams, TComOutputBitstream());
      pcSlice->clearSubstreamSizes();
      UInt numBinsCoded = 0;
      m_pcSliceEncoder->encodeSlice(pcPic, &substreamsOut[0], numBinsCoded);
    }
    if (bLF || bSAO) {
      hm355_planes rec; for (Int c = 0; c < 3; c++) rec.plane[c] = pcPic->getPicYuvRec()->getAddr(ComponentID(c));
      if (hm355_download(m_pcEncTop->getDeviceContext(), 0, &rec, NULL, NULL) != HM355_OK) { fprintf(stderr, "TEncGOP::compressGOP: download failed: %s\n", hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
    }
  }
}

// ---- loop filters: the picture is still resident in device slot 0 after TEncSlice::compressSlice ----
Void TComLoopFilter::loopFilterPic(TComPic *pcPic)
{
  hm355_dbk_desc dd; memset(&dd, 0, sizeof(dd));
  dd.slice_type = (int32_t)pcPic->getSlice(0)->getSliceType(); dd.qp = pcPic->getSlice(0)->getSliceQp();
  if (hm355_deblock_run(m_pcEncTop->getDeviceContext(), 1, &dd) != HM355_OK) { fprintf(stderr, "TComLoopFilter::loopFilterPic: %s\n", hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
}
Void TEncSampleAdaptiveOffset::SAOProcess(TComPic *pPic, Bool *sliceEnabled, const Double *lambdas)
{
  hm355_sao_desc sd; memset(&sd, 0, sizeof(sd));
  sd.qp = pPic->getSlice(0)->getSliceQp(); sd.cabac_init_type = (int32_t)pPic->getSlice(0)->getSliceType(); sd.depth = 0;   // all-intra GOP: temporal depth 0
  sd.lambda = lambdas[0]; sd.chroma_weight = lambdas[0] / lambdas[1];
  memcpy(sd.disabled_rate, m_saoDisabledRate, sizeof(m_saoDisabledRate));
  if (hm355_sao_run(m_pcEncTop->getDeviceContext(), 1, &sd) != HM355_OK) { fprintf(stderr, "TEncSampleAdaptiveOffset::SAOProcess: %s\n", hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
  memcpy(m_saoDisabledRate, sd.disabled_rate, sizeof(m_saoDisabledRate));
  for (Int c = 0; c < 3; c++) sliceEnabled[c] = sd.enabled[c] != 0;
}

// ---- TEncSlice ----
Void TEncSlice::init(TEncTop *t) { m_pcEncTop = t; }
Void TEncSlice::setUpLambda(TComSlice *, const Double dLambda, Int iQP)
{
  m_dLambda = dLambda;
  const Int q = iQP < 0 ? 0 : (iQP > 57 ? 57 : iQP);
  const Int qpc = kChromaScale420[q];                       // chroma QP offsets are 0 in every config
  m_dChromaWeight = pow(2.0, (iQP - qpc) / 3.0);
  m_dLambdas[0] = dLambda; m_dLambdas[1] = m_dLambdas[2] = dLambda / m_dChromaWeight;   // :150-155
}
Void TEncSlice::initEncSlice(TComPic *pcPic, Int, Int, Int, Int, TComSlice *&rpcSlice)
{
  rpcSlice = pcPic->getSlice(0);
  rpcSlice->setSliceType(I_SLICE);                          // IntraPeriod 1: every picture is an I slice
  const Double dQP = m_pcEncTop->getQP();
  const Int NumberBFrames = m_pcEncTop->getGOPSize() - 1;
  Double s = 0.05 * (Double)NumberBFrames; s = s < 0.0 ? 0.0 : (s > 0.5 ? 0.5 : s);
  const Double dLambda_scale = 1.0 - s;
  const Double dLambda = 0.57 * dLambda_scale * pow(2.0, (dQP - 12) / 3.0);   // TEncSlice.cpp:323-352
  const Int iQP = (Int)floor(dQP + 0.5);
  setUpLambda(rpcSlice, dLambda, iQP);
  rpcSlice->setSliceQp(iQP);
}
Void TEncSlice::encodeSlice(TComPic *pcPic, TComOutputBitstream *pcSubstreams, UInt &numBinsCoded)
{ // the picture (CU / TU data, coefficients, SAO parameters) is still resident in device slot 0
  TComSlice *pcSlice = pcPic->getSlice(0);
  hm355_ctx *ctx = m_pcEncTop->getDeviceContext();
  const Int numSubstreams = hm355_num_substreams(ctx);
  hm355_bits_desc bd; memset(&bd, 0, sizeof(bd));
  bd.slice_type = (int32_t)pcSlice->getSliceType(); bd.qp = pcSlice->getSliceQp(); bd.cabac_init_type = bd.slice_type; bd.max_merge_cand = 5;
  bd.sao_enabled[0] = pcSlice->getSaoEnabledFlag(0); bd.sao_enabled[1] = pcSlice->getSaoEnabledFlag(1);
  std::vector<uint8_t> bytes((size_t)m_pcEncTop->getSourceWidth() * m_pcEncTop->getSourceHeight() * 4 + 4096); std::vector<uint32_t> sizes(numSubstreams);
  bd.out = bytes.data(); bd.out_cap = bytes.size(); bd.sub_sizes = sizes.data();
  if (hm355_encode_slices_run(ctx, 1, &bd) != HM355_OK) { fprintf(stderr, "TEncSlice::encodeSlice: device path failed: %s\n", hm355_last_error(ctx)); exit(EXIT_FAILURE); }
  const uint8_t *p = bytes.data();
  for (Int k = 0; k < numSubstreams; p += sizes[k], k++) {
    pcSubstreams[k].getFIFO().assign(p, p + sizes[k]);
    if (k + 1 < numSubstreams) pcSlice->addSubstreamSize(sizes[k]);      // TEncSlice.cpp:1067-1071 (+ the start code emulation count, a NAL-level matter)
  }
  numBinsCoded = bd.num_bins;
}
Void TEncSlice::compressSlice(TComPic *pcPic)
{
  TComSlice *pcSlice = pcPic->getSlice(0);
  hm355_slice_desc sd; sd.slice_type = (int32_t)pcSlice->getSliceType(); sd.qp = pcSlice->getSliceQp(); sd.lambda = m_dLambda; sd.chroma_weight = m_dChromaWeight;
  hm355_planes org, rec;
  for (Int c = 0; c < 3; c++) { org.plane[c] = pcPic->getPicYuvOrg()->getAddr(ComponentID(c)); rec.plane[c] = pcPic->getPicYuvRec()->getAddr(ComponentID(c)); }
  hm355_slice_stats st;
  const int rc = hm355_compress_slice(m_pcEncTop->getDeviceContext(), &sd, &org, &rec, pcPic->getCtu(0), &st);
  if (rc != HM355_OK) { fprintf(stderr, "TEncSlice::compressSlice: device path failed (%d): %s\n", rc, hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
  m_uiPicTotalBits = st.pic_total_bits; m_dPicRdCost = st.pic_rd_cost; m_uiPicDist = st.pic_dist;     // TEncSlice.cpp:889-891
}
