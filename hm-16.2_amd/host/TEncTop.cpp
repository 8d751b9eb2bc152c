#include "TEncTop.h"
#include <algorithm>
#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const unsigned char kChromaScale420[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };

Void TComPic::create(Int w, Int h)
{
  m_org.create(w, h); m_rec.create(w, h);
  const size_t n = (size_t)((w + 63) / 64) * ((h + 63) / 64);
  m_ctus.assign(n, hm355_ctu_out()); m_ictus.assign(n, hm355_ctu_inter_out());
}

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
Void TEncTop::destroy() { for (auto p : m_cListPic) { if (p->getDeviceRef()) hm355_ref_release(m_ctx, p->getDeviceRef()); delete p; } m_cListPic.clear(); if (m_ctx) hm355_destroy(m_ctx); m_ctx = nullptr; }
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
    if (getUseAdaptiveQP()) { m_cPreanalyzer.init(this); m_cPreanalyzer.xPreanalyze(pic); }   // TEncTop.cpp:271-274
    m_cListPic.push_back(pic);
  }
  if (!m_iNumPicRcvd || (!flush && m_iPOCLast != 0 && m_iNumPicRcvd != m_iGOPSize && m_iGOPSize)) return;   // TEncTop.cpp:277: POC 0 goes alone
  m_cGOPEncoder.compressGOP(m_iPOCLast, m_iNumPicRcvd, m_cListPic);
  iNumEncoded = m_iNumPicRcvd; m_iNumPicRcvd = 0;
  rcListPicOut = m_cListPic;
}

// ---- TEncGOP ----
Void TEncGOP::init(TEncTop *t) { m_pcEncTop = t; m_pcSliceEncoder = t->getSliceEncoder(); m_pcLoopFilter = t->getLoopFilter(); m_pcSAO = t->getSAO(); }
Void TEncGOP::compressGOP(Int iPOCLast, Int iNumPicRcvd, std::list<TComPic *> &rcListPic)
{
  hm355_ctx *ctx = m_pcEncTop->getDeviceContext();
  const Int gopSize = iPOCLast == 0 ? 1 : m_pcEncTop->getGOPSize();                                    // xInitGOP, TEncGOP.cpp:1794
  for (Int iGOPid = 0; iGOPid < gopSize; iGOPid++) {
    const Int pocCurr = iPOCLast == 0 ? 0 : iPOCLast - iNumPicRcvd + m_pcEncTop->getGOPEntry(iGOPid).m_POC;   // :757-771
    if (pocCurr >= m_pcEncTop->getFramesToBeEncoded()) continue;
    TComPic *pcPic = nullptr;
    for (auto p : rcListPic) if (p->getPOC() == pocCurr && !p->getReconMark()) pcPic = p;
    if (!pcPic) continue;
    TComSlice *pcSlice = nullptr;
    m_pcSliceEncoder->initEncSlice(pcPic, iPOCLast, pocCurr, iNumPicRcvd, iGOPid, pcSlice);            // :783
    if (pcSlice->getSliceType() == B_SLICE && m_pcEncTop->getGOPEntry(iGOPid).m_sliceType == 'P') pcSlice->setSliceType(P_SLICE);   // :800
    if (!pcSlice->isIntra()) xSetReferences(pcSlice, pocCurr, iGOPid, rcListPic);                      // :851-959
    if (pcSlice->getSliceType() == B_SLICE && pcSlice->getNumRefIdx(REF_PIC_LIST_1) == 0) pcSlice->setSliceType(P_SLICE);   // :961
    { // collocated picture of a B slice (:644-690, :967-996): from list 1 unless the closest following reference has a lower QP offset than
      // the closest preceding one; checkLDC when no reference follows the picture
      const GOPEntry &ge = m_pcEncTop->getGOPEntry(iGOPid);
      UInt uiColDir = 1; Int iCloseLeft = 1, iCloseRight = -1;
      for (Int i = 0; i < ge.m_numRefPics; i++) {
        const Int iRef = ge.m_referencePics[i];
        if (iRef > 0 && (iRef < iCloseRight || iCloseRight == -1)) iCloseRight = iRef;
        else if (iRef < 0 && (iRef > iCloseLeft || iCloseLeft == 1)) iCloseLeft = iRef;
      }
      if (iCloseRight > -1) iCloseRight = iCloseRight + ge.m_POC - 1;
      if (iCloseLeft < 1) { iCloseLeft = iCloseLeft + ge.m_POC - 1; while (iCloseLeft < 0) iCloseLeft += gopSize; }
      Int iLeftQP = 0, iRightQP = 0;
      for (Int i = 0; i < gopSize; i++) {
        if (m_pcEncTop->getGOPEntry(i).m_POC == (iCloseLeft % gopSize) + 1) iLeftQP = m_pcEncTop->getGOPEntry(i).m_QPOffset;
        if (iCloseRight > -1 && m_pcEncTop->getGOPEntry(i).m_POC == (iCloseRight % gopSize) + 1) iRightQP = m_pcEncTop->getGOPEntry(i).m_QPOffset;
      }
      if (iCloseRight > -1 && iRightQP < iLeftQP) uiColDir = 0;
      pcSlice->setColFromL0Flag(pcSlice->getSliceType() == B_SLICE ? 1 - uiColDir : 1); pcSlice->setColRefIdx(0);
      Bool bLowDelay = true;
      for (Int l = 0; l < 2; l++) for (Int i = 0; i < pcSlice->getNumRefIdx(RefPicList(l)); i++) if (pcSlice->getRefPOC(RefPicList(l), i) > pocCurr) bLowDelay = false;
      pcSlice->setCheckLDC(pcSlice->getSliceType() == B_SLICE ? bLowDelay : true);
    }
    pcSlice->setEnableTMVPFlag(m_pcEncTop->getTMVPModeId() == 1);                                     // :1017-1025
    { // mvd_l1_zero_flag when both lists hold the same pictures in the same order (:1027-1058)
      Bool same = pcSlice->getSliceType() == B_SLICE && pcSlice->getNumRefIdx(REF_PIC_LIST_0) == pcSlice->getNumRefIdx(REF_PIC_LIST_1);
      for (Int i = 0; same && i < pcSlice->getNumRefIdx(REF_PIC_LIST_1); i++) same = pcSlice->getRefPOC(REF_PIC_LIST_1, i) == pcSlice->getRefPOC(REF_PIC_LIST_0, i);
      pcSlice->setMvdL1ZeroFlag(same);
    }
    pcSlice->setMaxNumMergeCand(m_pcEncTop->getMaxNumMergeCand());
    { // context table of the slice: TEncSbac::resetEntropy :106-115 with cabac_init_present_flag (TEncTop::xInitPPS)
      const Int idx = m_pcEncTop->getEncCABACTableIdx();
      pcSlice->setCabacInitType((!pcSlice->isIntra() && (idx == B_SLICE || idx == P_SLICE)) ? idx : (Int)pcSlice->getSliceType());
    }
    m_pcSliceEncoder->precompressSlice(pcPic);                                                        // :1137
    const auto tPic0 = std::chrono::steady_clock::now();
    m_pcSliceEncoder->compressSlice(pcPic);                                                           // :1138
    double searchKernelMs = 0; hm355_last_run_info(ctx, &searchKernelMs, NULL);
    const double searchWallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tPic0).count();
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
      const Int numSubstreams = hm355_num_substreams(ctx);
      std::vector<TComOutputBitstream> &substreamsOut = pcPic->getSubstreams();
      substreamsOut.assign(numSubstreams, TComOutputBitstream());
      pcSlice->clearSubstreamSizes();
      UInt numBinsCoded = 0;
      m_pcSliceEncoder->encodeSlice(pcPic, &substreamsOut[0], numBinsCoded);
    }
    if (bLF || bSAO) {
      hm355_planes rec; for (Int c = 0; c < 3; c++) rec.plane[c] = pcPic->getPicYuvRec()->getAddr(ComponentID(c));
      if (hm355_download(ctx, 0, &rec, NULL, NULL) != HM355_OK) { fprintf(stderr, "TEncGOP::compressGOP: download failed: %s\n", hm355_last_error(ctx)); exit(EXIT_FAILURE); }
    }
    pcPic->setReconMark(true); m_codedPics.push_back(pcPic);
    if (getenv("HM355_TIMING"))   // one line per picture for bench.py: the search (TEncSlice::compressSlice) as the HIP events and the host clock saw it
      fprintf(stderr, "{\"poc\": %d, \"slice_type\": %d, \"qp\": %d, \"search_kernel_ms\": %.3f, \"search_wall_ms\": %.3f, \"picture_wall_ms\": %.3f, \"bits\": %llu}\n", pocCurr,
              (int)pcSlice->getSliceType(), pcSlice->getSliceQp(), searchKernelMs, searchWallMs,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tPic0).count(), (unsigned long long)m_pcSliceEncoder->getTotalBits());
    if (m_pcEncTop->getGOPSize() > 1) {
      // the finished picture becomes a reference on the device: border extension + TComPic::compressMotion (:1660), no host round trip
      int32_t numRef[2] = { pcSlice->getNumRefIdx(REF_PIC_LIST_0), pcSlice->getNumRefIdx(REF_PIC_LIST_1) }, refPoc[2][16], refLT[2][16];
      memset(refPoc, 0, sizeof(refPoc)); memset(refLT, 0, sizeof(refLT));
      for (Int l = 0; l < 2; l++) for (Int i = 0; i < numRef[l]; i++) refPoc[l][i] = pcSlice->getRefPOC(RefPicList(l), i);
      hm355_ref *ref = nullptr;
      if (hm355_ref_from_slot(ctx, 0, pocCurr, pcSlice->isIntra() ? 0 : 1, numRef, refPoc, refLT, &ref) != HM355_OK) { fprintf(stderr, "TEncGOP::compressGOP: %s\n", hm355_last_error(ctx)); exit(EXIT_FAILURE); }
      pcPic->setDeviceRef(ref);
      for (auto p : rcListPic) if (p->getDeviceRef() && p->getPOC() < pocCurr - 24) { hm355_ref_release(ctx, p->getDeviceRef()); p->setDeviceRef(nullptr); }   // out of every reference picture set
    }
  }
}

Void TEncGOP::xSetReferences(TComSlice *pcSlice, Int pocCurr, Int iGOPid, std::list<TComPic *> &rcListPic)
{
  const GOPEntry &ge = m_pcEncTop->getGOPEntry(iGOPid);
  auto find = [&](Int poc) -> TComPic * { for (auto p : rcListPic) if (p->getPOC() == poc && p->getReconMark() && p->getDeviceRef()) return p; return nullptr; };
  std::vector<Int> refs; Bool missing = false;
  for (Int i = 0; i < ge.m_numRefPics; i++) { const Int poc = pocCurr + ge.m_referencePics[i]; if (poc >= 0 && find(poc)) refs.push_back(poc); else missing = true; }
  // start of the sequence (TAppEncCfg.cpp xCheckParameter, the extra reference picture sets): pictures before POC 0 are replaced by the most
  // recently coded ones of the same or a lower temporal layer, stepping backwards in coding order, up to the number of active references
  if (missing)
    for (Int k = (Int)m_codedPics.size() - 1; k >= 0 && (Int)refs.size() < ge.m_numRefPicsActive; k--) {
      TComPic *p = m_codedPics[k];
      Bool have = false; for (Int r : refs) have = have || r == p->getPOC();
      if (!have && p->getDeviceRef() && p->getTLayer() <= ge.m_temporalId) refs.push_back(p->getPOC());
    }
  // TComSlice::setRefPicList (no list modification): list 0 = the pictures before the current one, closest first, then the ones after it, closest
  // first; list 1 the other way round
  std::vector<Int> before, after;
  for (Int r : refs) (r < pocCurr ? before : after).push_back(r);
  for (size_t i = 0; i < before.size(); i++) for (size_t j = i + 1; j < before.size(); j++) if (before[j] > before[i]) { const Int t = before[i]; before[i] = before[j]; before[j] = t; }
  for (size_t i = 0; i < after.size(); i++) for (size_t j = i + 1; j < after.size(); j++) if (after[j] < after[i]) { const Int t = after[i]; after[i] = after[j]; after[j] = t; }
  std::vector<Int> l0(before), l1(after);
  l0.insert(l0.end(), after.begin(), after.end()); l1.insert(l1.end(), before.begin(), before.end());
  const Int n = (Int)refs.size() < ge.m_numRefPicsActive ? (Int)refs.size() : ge.m_numRefPicsActive;     // TEncGOP.cpp:951-952
  const Bool isB = pcSlice->getSliceType() == B_SLICE;
  pcSlice->setNumRefIdx(REF_PIC_LIST_0, n); pcSlice->setNumRefIdx(REF_PIC_LIST_1, isB ? n : 0);
  for (Int i = 0; i < n; i++) {
    pcSlice->setRefPic(find(l0[i]), REF_PIC_LIST_0, i); pcSlice->setRefPOC(l0[i], REF_PIC_LIST_0, i);
    if (isB) { pcSlice->setRefPic(find(l1[i]), REF_PIC_LIST_1, i); pcSlice->setRefPOC(l1[i], REF_PIC_LIST_1, i); }
  }
}

// ---- loop filters: the picture is still resident in device slot 0 after TEncSlice::compressSlice ----
Void TComLoopFilter::loopFilterPic(TComPic *pcPic)
{
  hm355_dbk_desc dd; memset(&dd, 0, sizeof(dd));
  dd.slice_type = (int32_t)pcPic->getSlice(0)->getSliceType(); dd.qp = pcPic->getSlice(0)->getSliceQp();
  for (Int l = 0; l < 2; l++) for (Int i = 0; i < pcPic->getSlice(0)->getNumRefIdx(RefPicList(l)); i++) dd.ref_poc[l][i] = pcPic->getSlice(0)->getRefPOC(RefPicList(l), i);
  if (hm355_deblock_run(m_pcEncTop->getDeviceContext(), 1, &dd) != HM355_OK) { fprintf(stderr, "TComLoopFilter::loopFilterPic: %s\n", hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
}
Void TEncSampleAdaptiveOffset::SAOProcess(TComPic *pPic, Bool *sliceEnabled, const Double *lambdas)
{
  hm355_sao_desc sd; memset(&sd, 0, sizeof(sd));
  sd.qp = pPic->getSlice(0)->getSliceQp(); sd.cabac_init_type = pPic->getSlice(0)->getCabacInitType(); sd.depth = pPic->getSlice(0)->getDepth();
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
Void TEncSlice::initEncSlice(TComPic *pcPic, Int pocLast, Int pocCurr, Int, Int iGOPid, TComSlice *&rpcSlice)
{
  rpcSlice = pcPic->getSlice(0);
  const Int gopSize = m_pcEncTop->getGOPSize();
  const GOPEntry &ge = m_pcEncTop->getGOPEntry(iGOPid);
  // depth of the picture in the GOP hierarchy (TEncSlice.cpp:196-231)
  Int depth = 0;
  { Int poc = rpcSlice->getPOC() % gopSize;
    if (poc != 0) {
      Int step = gopSize;
      for (Int i = step >> 1; i >= 1; i >>= 1) {
        for (Int j = i; j < gopSize; j += step) if (j == poc) { i = 0; break; }
        step >>= 1; depth++;
      }
    } }
  // slice type (:245-262): POC 0 and every IntraPeriod-th picture are I slices (the period is unsigned: -1 never matches)
  SliceType eSliceType = (pocLast == 0 || (UInt)pocCurr % (UInt)m_pcEncTop->getIntraPeriod() == 0 || gopSize == 0) ? I_SLICE : B_SLICE;
  rpcSlice->setSliceType(eSliceType);
  // QP and lambda (:287-352): no delta QP, no lossless mode, lambda modifiers 1
  Double dQP = m_pcEncTop->getQP();
  if (eSliceType != I_SLICE) dQP += ge.m_QPOffset;
  const Int NumberBFrames = gopSize - 1;
  Double sc = 0.05 * (Double)NumberBFrames; sc = sc < 0.0 ? 0.0 : (sc > 0.5 ? 0.5 : sc);
  const Double dLambda_scale = 1.0 - sc;
  const Double qp_temp = dQP - 12;                                          // bitdepth_luma_qp_scale = 0 (TEncSlice.cpp:322)
  Double dQPFactor = ge.m_QPFactor;
  if (eSliceType == I_SLICE) dQPFactor = 0.57 * dLambda_scale;
  Double dLambda = dQPFactor * pow(2.0, qp_temp / 3.0);
  if (depth > 0) { Double c = qp_temp / 6.0; c = c < 2.0 ? 2.0 : (c > 4.0 ? 4.0 : c); dLambda *= c; }
  if (!m_pcEncTop->getUseHADME() && eSliceType != I_SLICE) dLambda *= 0.95;
  Int iQP = (Int)floor(dQP + 0.5); iQP = iQP > 51 ? 51 : (iQP < 0 ? 0 : iQP);
  setUpLambda(rpcSlice, dLambda, iQP);
  rpcSlice->setSliceQp(iQP); rpcSlice->setLambda(dLambda);
  rpcSlice->setNumRefIdx(REF_PIC_LIST_0, 0); rpcSlice->setNumRefIdx(REF_PIC_LIST_1, 0);
  rpcSlice->setDepth(depth);
  pcPic->setTLayer(eSliceType == I_SLICE ? 0 : ge.m_temporalId);                // :457-462
}
Void TEncSlice::encodeSlice(TComPic *pcPic, TComOutputBitstream *pcSubstreams, UInt &numBinsCoded)
{ // the picture (CU / TU data, coefficients, SAO parameters) is still resident in device slot 0
  TComSlice *pcSlice = pcPic->getSlice(0);
  hm355_ctx *ctx = m_pcEncTop->getDeviceContext();
  const Int numSubstreams = hm355_num_substreams(ctx);
  hm355_bits_desc bd; memset(&bd, 0, sizeof(bd));
  bd.slice_type = (int32_t)pcSlice->getSliceType(); bd.qp = pcSlice->getSliceQp(); bd.cabac_init_type = pcSlice->getCabacInitType();
  bd.max_merge_cand = (int32_t)pcSlice->getMaxNumMergeCand(); bd.mvd_l1_zero = pcSlice->getMvdL1ZeroFlag();
  bd.num_ref_idx[0] = pcSlice->getNumRefIdx(REF_PIC_LIST_0); bd.num_ref_idx[1] = pcSlice->getNumRefIdx(REF_PIC_LIST_1);
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
  m_pcEncTop->setEncCABACTableIdx(bd.next_cabac_init_type);               // determineCabacInitIdx, TEncSlice.cpp:1083-1093 (cabac_init_present_flag)
}
// TEncPreanalyzer.cpp:64-139 for the CTU-sized units of layer 0
Void TEncPreanalyzer::xPreanalyze(TComPic *pcPic)
{
  hm355_ctx *ctx = m_pcEncTop->getDeviceContext();
  hm355_planes org; for (Int c = 0; c < 3; c++) org.plane[c] = pcPic->getPicYuvOrg()->getAddr(ComponentID(c));
  const UInt n = pcPic->getNumberOfCtusInFrame();
  std::vector<uint64_t> sums((size_t)n * 8);
  if (hm355_upload(ctx, 0, &org) != HM355_OK || hm355_preanalyze(ctx, 0, sums.data()) != HM355_OK) { fprintf(stderr, "TEncPreanalyzer::xPreanalyze: %s\n", hm355_last_error(ctx)); exit(EXIT_FAILURE); }
  const Int w = m_pcEncTop->getSourceWidth(), h = m_pcEncTop->getSourceHeight(), wCtu = (w + 63) / 64;
  std::vector<Double> &act = pcPic->getAQActivities(); act.assign(n, 0.0);
  Double dSumAct = 0.0;
  for (UInt a = 0; a < n; a++) {
    const Int cw = std::min(64, w - (Int)(a % wCtu) * 64), ch = std::min(64, h - (Int)(a / wCtu) * 64);
    const UInt uiNumPixInAQPart = (UInt)(cw * ch);
    Double dMinVar = 1.7976931348623157e308;
    for (Int i = 0; i < 4; i++) {
      const Double dAverage = Double(sums[(size_t)a * 8 + i]) / uiNumPixInAQPart;
      const Double dVariance = Double(sums[(size_t)a * 8 + 4 + i]) / uiNumPixInAQPart - dAverage * dAverage;
      dMinVar = std::min(dMinVar, dVariance);
    }
    act[a] = 1.0 + dMinVar;
    dSumAct += act[a];
  }
  pcPic->setAvgActivity(dSumAct / n);
}
Int TEncSlice::xComputeQP(TComPic *pcPic, UInt ctuRsAddr, Int sliceQp)
{
  const Double dMaxQScale = pow(2.0, m_pcEncTop->getQPAdaptationRange() / 6.0);
  const Double dAvgAct = pcPic->getAvgActivity(), dCUAct = pcPic->getAQActivities()[ctuRsAddr];
  const Double dNormAct = (dMaxQScale * dCUAct + dAvgAct) / (dCUAct + dMaxQScale * dAvgAct);
  const Double dQpOffset = log(dNormAct) / log(2.0) * 6.0;
  const Int iQpOffset = Int(floor(dQpOffset + 0.49999));
  return std::min(51, std::max(-6 * (m_pcEncTop->getInternalBitDepth() - 8), sliceQp + iQpOffset));
}
Void TEncSlice::compressSlice(TComPic *pcPic)
{
  TComSlice *pcSlice = pcPic->getSlice(0);
  if (m_pcEncTop->getUseAdaptiveQP()) { // cu_qp_delta is on (TEncTop::xInitPPS :608-622): every CTU at its xComputeQP, TEncCu::m_bEncodeDQP handed in
    std::vector<int8_t> qp(pcPic->getNumberOfCtusInFrame());
    for (UInt a = 0; a < pcPic->getNumberOfCtusInFrame(); a++) qp[a] = (int8_t)xComputeQP(pcPic, a, pcSlice->getSliceQp());
    hm355_dqp_desc dq; dq.use_dqp = 1; dq.dqp_flag_in = m_bEncodeDQP ? 1 : 0; dq.ctu_qp = qp.data();
    if (hm355_set_dqp(m_pcEncTop->getDeviceContext(), 0, &dq) != HM355_OK) { fprintf(stderr, "TEncSlice::compressSlice: %s\n", hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
  }
  if (!pcSlice->isIntra()) { // P / B slice: the reference pictures are device-resident (hm355_ref_from_slot)
    hm355_inter_slice_desc d; memset(&d, 0, sizeof(d));
    d.base.slice_type = (int32_t)pcSlice->getSliceType(); d.base.qp = pcSlice->getSliceQp(); d.base.lambda = m_dLambda; d.base.chroma_weight = m_dChromaWeight;
    d.poc = pcSlice->getPOC(); d.cabac_init_type = pcSlice->getCabacInitType();
    for (Int l = 0; l < 2; l++) {
      d.num_ref_idx[l] = pcSlice->getNumRefIdx(RefPicList(l));
      for (Int i = 0; i < d.num_ref_idx[l]; i++) d.dev_ref[l][i] = pcSlice->getRefPic(RefPicList(l), i)->getDeviceRef();
    }
    d.col_from_l0 = (int32_t)pcSlice->getColFromL0Flag(); d.col_ref_idx = (int32_t)pcSlice->getColRefIdx(); d.tmvp = pcSlice->getEnableTMVPFlag();
    d.mvd_l1_zero = pcSlice->getMvdL1ZeroFlag(); d.max_merge_cand = (int32_t)pcSlice->getMaxNumMergeCand(); d.check_ldc = pcSlice->getCheckLDC();
    d.lambda_motion_sad = (uint32_t)floor(65536.0 * sqrt(m_dLambda)); d.lambda_motion_sse = (uint32_t)floor(65536.0 * m_dLambda);   // TComRdCost::setLambda, TComRdCost.cpp:194-218
    hm355_planes org, rec;
    for (Int c = 0; c < 3; c++) { org.plane[c] = pcPic->getPicYuvOrg()->getAddr(ComponentID(c)); rec.plane[c] = pcPic->getPicYuvRec()->getAddr(ComponentID(c)); }
    hm355_slice_stats st;
    const int rc = hm355_compress_slice_inter(m_pcEncTop->getDeviceContext(), &d, &org, &rec, pcPic->getCtu(0), pcPic->getCtuInter(0), &st);
    if (rc != HM355_OK) { fprintf(stderr, "TEncSlice::compressSlice: device path failed (%d): %s\n", rc, hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
    m_uiPicTotalBits = st.pic_total_bits; m_dPicRdCost = st.pic_rd_cost; m_uiPicDist = st.pic_dist;
    if (m_pcEncTop->getUseAdaptiveQP()) { int32_t f = 0; hm355_get_dqp(m_pcEncTop->getDeviceContext(), 0, NULL, &f); m_bEncodeDQP = f != 0; }
    return;
  }
  hm355_slice_desc sd; sd.slice_type = (int32_t)pcSlice->getSliceType(); sd.qp = pcSlice->getSliceQp(); sd.lambda = m_dLambda; sd.chroma_weight = m_dChromaWeight;
  hm355_planes org, rec;
  for (Int c = 0; c < 3; c++) { org.plane[c] = pcPic->getPicYuvOrg()->getAddr(ComponentID(c)); rec.plane[c] = pcPic->getPicYuvRec()->getAddr(ComponentID(c)); }
  hm355_slice_stats st;
  const int rc = hm355_compress_slice(m_pcEncTop->getDeviceContext(), &sd, &org, &rec, pcPic->getCtu(0), &st);
  if (rc != HM355_OK) { fprintf(stderr, "TEncSlice::compressSlice: device path failed (%d): %s\n", rc, hm355_last_error(m_pcEncTop->getDeviceContext())); exit(EXIT_FAILURE); }
  m_uiPicTotalBits = st.pic_total_bits; m_dPicRdCost = st.pic_rd_cost; m_uiPicDist = st.pic_dist;     // TEncSlice.cpp:889-891
  if (m_pcEncTop->getUseAdaptiveQP()) { int32_t f = 0; hm355_get_dqp(m_pcEncTop->getDeviceContext(), 0, NULL, &f); m_bEncodeDQP = f != 0; }
}
