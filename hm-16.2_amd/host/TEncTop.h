// Host-side C++ mirror of the reference's encoder object graph for the hot path:
//   TEncTop  (source/Lib/TLibEncoder/TEncTop.h:70-176)   create / init / encode / destroy
//   TEncGOP  (TEncGOP.h; compressGOP TEncGOP.cpp:527)    per-picture driver
//   TEncSlice(TEncSlice.h:60-127)                        initEncSlice / setUpLambda / compressSlice
// Same class and method names, argument meaning and call order as the reference; the bodies are ours and
// TEncSlice::compressSlice forwards to the C ABI (include/hm355.h) instead of running the CPU search.
//   TComLoopFilter (TLibCommon/TComLoopFilter.h)              loopFilterPic          -> hm355_deblock_run
//   TEncSampleAdaptiveOffset (TEncSampleAdaptiveOffset.h)     initRDOCabacCoder / SAOProcess -> hm355_sao_run
// Everything else the reference does outside the hot path (RPS, SEI, NAL, entropy pass, rate control) is out of scope
//  and absent here.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <list>
#include <vector>
#include "../../include/hm355.h"

typedef void Void; typedef bool Bool; typedef int Int; typedef unsigned int UInt; typedef double Double;
typedef short Pel;
enum SliceType { B_SLICE = 0, P_SLICE = 1, I_SLICE = 2 };
enum ComponentID { COMPONENT_Y = 0, COMPONENT_Cb = 1, COMPONENT_Cr = 2, MAX_NUM_COMPONENT = 3 };

// picture planes (TComPicYuv.h): tightly packed 16-bit samples, 4:2:0
class TComPicYuv {
public:
  Void create(Int w, Int h) { m_w = w; m_h = h; for (Int c = 0; c < 3; c++) m_buf[c].assign((size_t)(w >> (c ? 1 : 0)) * (h >> (c ? 1 : 0)), 0); }
  Void destroy() { for (Int c = 0; c < 3; c++) m_buf[c].clear(); }
  uint16_t *getAddr(ComponentID c) { return m_buf[c].data(); }
  Int getWidth(ComponentID c) const { return m_w >> (c ? 1 : 0); }
  Int getHeight(ComponentID c) const { return m_h >> (c ? 1 : 0); }
private:
  Int m_w = 0, m_h = 0; std::vector<uint16_t> m_buf[3];
};

class TComPic;
enum RefPicList { REF_PIC_LIST_0 = 0, REF_PIC_LIST_1 = 1 };
// slice header fields compressSlice reads (TComSlice.h)
class TComSlice {
public:
  // inter slices: what TEncGOP::compressGOP sets up between initEncSlice and compressSlice (TEncGOP.cpp:851-1058)
  Void setDepth(Int d) { m_depth = d; } Int getDepth() const { return m_depth; }
  Void setNumRefIdx(RefPicList e, Int n) { m_numRefIdx[e] = n; } Int getNumRefIdx(RefPicList e) const { return m_numRefIdx[e]; }
  Void setRefPic(TComPic *p, RefPicList e, Int i) { m_refPic[e][i] = p; } TComPic *getRefPic(RefPicList e, Int i) const { return m_refPic[e][i]; }
  Void setRefPOC(Int poc, RefPicList e, Int i) { m_refPOC[e][i] = poc; } Int getRefPOC(RefPicList e, Int i) const { return m_refPOC[e][i]; }
  Bool isIntra() const { return m_type == I_SLICE; }
  Void setColFromL0Flag(UInt f) { m_colFromL0 = f; } UInt getColFromL0Flag() const { return m_colFromL0; }
  Void setColRefIdx(UInt i) { m_colRefIdx = i; } UInt getColRefIdx() const { return m_colRefIdx; }
  Void setEnableTMVPFlag(Bool b) { m_tmvp = b; } Bool getEnableTMVPFlag() const { return m_tmvp; }
  Void setMvdL1ZeroFlag(Bool b) { m_mvdL1Zero = b; } Bool getMvdL1ZeroFlag() const { return m_mvdL1Zero; }
  Void setMaxNumMergeCand(UInt n) { m_maxMergeCand = n; } UInt getMaxNumMergeCand() const { return m_maxMergeCand; }
  Void setCheckLDC(Bool b) { m_checkLDC = b; } Bool getCheckLDC() const { return m_checkLDC; }
  Void setCabacInitType(Int t) { m_cabacInitType = t; } Int getCabacInitType() const { return m_cabacInitType; }   // table TEncSbac::resetEntropy uses (:106-115)
  Void setLambda(Double l) { m_lambda = l; } Double getLambda() const { return m_lambda; }
  Void setSliceType(SliceType t) { m_type = t; } SliceType getSliceType() const { return m_type; }
  Void setSliceQp(Int qp) { m_qp = qp; } Int getSliceQp() const { return m_qp; }
  Void setPOC(Int p) { m_poc = p; } Int getPOC() const { return m_poc; }
  Void setSliceBits(UInt b) { m_bits = b; } UInt getSliceBits() const { return m_bits; }
  Void setSaoEnabledFlag(Int chType, Bool b) { m_sao[chType] = b; } Bool getSaoEnabledFlag(Int chType) const { return m_sao[chType]; }
  Void clearSubstreamSizes() { m_substreamSizes.clear(); } Void addSubstreamSize(UInt s) { m_substreamSizes.push_back(s); }
  UInt getNumberOfSubstreamSizes() const { return (UInt)m_substreamSizes.size(); } UInt getSubstreamSize(Int i) const { return m_substreamSizes[i]; }
private:
  SliceType m_type = I_SLICE; Int m_qp = 32, m_poc = 0; UInt m_bits = 0; Bool m_sao[2] = {false, false}; std::vector<UInt> m_substreamSizes;
  Int m_depth = 0, m_numRefIdx[2] = {0, 0}, m_refPOC[2][16] = {}, m_cabacInitType = 2; TComPic *m_refPic[2][16] = {};
  UInt m_colFromL0 = 1, m_colRefIdx = 0, m_maxMergeCand = 5; Bool m_tmvp = true, m_mvdL1Zero = false, m_checkLDC = true; Double m_lambda = 0;
};

// byte FIFO of one substream (TComBitStream.h:89-160): only what encodeSlice's callers read
class TComOutputBitstream {
public:
  std::vector<uint8_t> &getFIFO() { return m_fifo; }
  const std::vector<uint8_t> &getFIFO() const { return m_fifo; }
  UInt getByteStreamLength() const { return (UInt)m_fifo.size(); }
  UInt getNumberOfWrittenBits() const { return (UInt)m_fifo.size() * 8; }
  Void clear() { m_fifo.clear(); }
private:
  std::vector<uint8_t> m_fifo;
};

// picture = original + reconstruction + per-CTU decision data (TComPic.h / TComPicSym.h / TComDataCU.h)
class TComPic {
public:
  Void create(Int w, Int h);
  TComPicYuv *getPicYuvOrg() { return &m_org; }
  TComPicYuv *getPicYuvRec() { return &m_rec; }
  TComSlice *getSlice(Int) { return &m_slice; }
  hm355_ctu_out *getCtu(UInt ctuRsAddr) { return &m_ctus[ctuRsAddr]; }     // the arrays of TComDataCU
  UInt getNumberOfCtusInFrame() const { return (UInt)m_ctus.size(); }
  Int getPOC() { return m_slice.getPOC(); }
  std::vector<TComOutputBitstream> &getSubstreams() { return m_substreams; }   // the slice data of the picture (kept where TEncGOP would hand it to the NAL writer)
  hm355_ctu_inter_out *getCtuInter(UInt ctuRsAddr) { return &m_ictus[ctuRsAddr]; }   // m_acCUMvField / merge / skip arrays of TComDataCU (inter slices)
  Void setReconMark(Bool b) { m_reconMark = b; } Bool getReconMark() const { return m_reconMark; }
  Void setTLayer(Int t) { m_tLayer = t; } Int getTLayer() const { return m_tLayer; }
  Void setDeviceRef(hm355_ref *r) { m_devRef = r; } hm355_ref *getDeviceRef() const { return m_devRef; }   // the finished picture as a device-resident reference
  // TEncPic (TEncPic.h:100-112), layer 0 of the QP adaptation: one activity per CTU and their average (TEncPreanalyzer::xPreanalyze)
  std::vector<Double> &getAQActivities() { return m_aqAct; } Void setAvgActivity(Double d) { m_aqAvg = d; } Double getAvgActivity() const { return m_aqAvg; }
private:
  std::vector<Double> m_aqAct; Double m_aqAvg = 0;
  TComPicYuv m_org, m_rec; TComSlice m_slice; std::vector<hm355_ctu_out> m_ctus; std::vector<TComOutputBitstream> m_substreams;
  std::vector<hm355_ctu_inter_out> m_ictus; Bool m_reconMark = false; hm355_ref *m_devRef = nullptr; Int m_tLayer = 0;
};

// one line of the GOP table of a cfg file (TEncCfg.h GOPEntry): "Frame1: P 1 3 0.4624 0 0 0 4 4 -1 -5 -9 -13 0"
struct GOPEntry {
  Int m_POC = 1, m_QPOffset = 0; Double m_QPFactor = 0.57; Int m_temporalId = 0, m_numRefPicsActive = 0, m_numRefPics = 0, m_referencePics[16] = {}; char m_sliceType = 'I';
};
// configuration holder (TEncCfg.h): the subset the hot path reads; unsupported values are rejected by create()
class TEncCfg {
public:
  Void setGOPEntry(Int i, const GOPEntry &e) { m_GOPList[i] = e; } const GOPEntry &getGOPEntry(Int i) const { return m_GOPList[i]; }
  Void setUseHADME(Bool b) { m_bUseHADME = b; } Bool getUseHADME() const { return m_bUseHADME; }
  Void setMaxNumMergeCand(UInt n) { m_maxNumMergeCand = n; } UInt getMaxNumMergeCand() const { return m_maxNumMergeCand; }
  Void setTMVPModeId(Int m) { m_TMVPModeId = m; } Int getTMVPModeId() const { return m_TMVPModeId; }
  Int getFramesToBeEncoded() const { return m_framesToBeEncoded; }
  Void setUseAdaptiveQP(Bool b) { m_bUseAdaptiveQP = b; } Bool getUseAdaptiveQP() const { return m_bUseAdaptiveQP; }                  // --AdaptiveQP
  Void setQPAdaptationRange(Int r) { m_iQPAdaptationRange = r; } Int getQPAdaptationRange() const { return m_iQPAdaptationRange; }   // --MaxQPAdaptationRange
  Void setSourceWidth(Int v) { m_iSourceWidth = v; } Void setSourceHeight(Int v) { m_iSourceHeight = v; }
  Void setInternalBitDepth(Int v) { m_bitDepth = v; } Void setQP(Int v) { m_iQP = v; }
  Void setIntraPeriod(Int v) { m_uiIntraPeriod = v; } Void setGOPSize(Int v) { m_iGOPSize = v; }
  Void setWaveFrontSynchro(Int v) { m_iWaveFrontSynchro = v; } Void setFramesToBeEncoded(Int v) { m_framesToBeEncoded = v; }
  Void setLoopFilterDisable(Bool b) { m_bLoopFilterDisable = b; } Bool getLoopFilterDisable() const { return m_bLoopFilterDisable; }
  Void setUseSAO(Bool b) { m_bUseSAO = b; } Bool getUseSAO() const { return m_bUseSAO; }
  Int getSourceWidth() const { return m_iSourceWidth; } Int getSourceHeight() const { return m_iSourceHeight; }
  Int getQP() const { return m_iQP; } Int getGOPSize() const { return m_iGOPSize; } Int getIntraPeriod() const { return m_uiIntraPeriod; }
  Int getWaveFrontsynchro() const { return m_iWaveFrontSynchro; } Int getInternalBitDepth() const { return m_bitDepth; }
protected:
  Int m_iSourceWidth = 0, m_iSourceHeight = 0, m_bitDepth = 8, m_iQP = 32, m_uiIntraPeriod = 1, m_iGOPSize = 1, m_iWaveFrontSynchro = 0, m_framesToBeEncoded = 0;
  Bool m_bLoopFilterDisable = true, m_bUseSAO = false;      // the loop filters are opt-in here (the reference's cfg files switch both on)
  Bool m_bUseAdaptiveQP = false; Int m_iQPAdaptationRange = 6;
  GOPEntry m_GOPList[16]; Bool m_bUseHADME = true; UInt m_maxNumMergeCand = 5; Int m_TMVPModeId = 1;
};

class TEncTop;

class TEncSlice {
public:
  Void init(TEncTop *pcEncTop);
  // TEncSlice::initEncSlice (TEncSlice.cpp:180-481): slice type, QP and lambda of the picture
  Void initEncSlice(TComPic *pcPic, Int pocLast, Int pocCurr, Int iNumPicRcvd, Int iGOPid, TComSlice *&rpcSlice);
  Void setUpLambda(TComSlice *slice, const Double dLambda, Int iQP);                 // TEncSlice.cpp:132-159
  const Double *getLambdas() const { return m_dLambdas; }                            // TComSlice::getLambdas of the current slice
  Void precompressSlice(TComPic *) {}                                                // DeltaQpRD = 0 in every config: no-op
  Void compressSlice(TComPic *pcPic);                                                // TEncSlice.cpp:640 -> hm355_compress_slice
  Int xComputeQP(TComPic *pcPic, UInt ctuRsAddr, Int sliceQp);                       // TEncCu::xComputeQP, TEncCu.cpp:1154 (the CTU-level unit of MaxCuDQPDepth 0)
  Void setdQPFlag(Bool b) { m_bEncodeDQP = b; } Bool getdQPFlag() const { return m_bEncodeDQP; }   // TEncCu::m_bEncodeDQP: carried from picture to picture
  Void encodeSlice(TComPic *pcPic, TComOutputBitstream *pcSubstreams, UInt &numBinsCoded);   // TEncSlice.cpp:910 -> hm355_encode_slices_run
  uint64_t getTotalBits() const { return m_uiPicTotalBits; }
  Double getPicRdCost() const { return m_dPicRdCost; }
  uint64_t getPicDist() const { return m_uiPicDist; }
private:
  TEncTop *m_pcEncTop = nullptr;
  Double m_dLambda = 0, m_dChromaWeight = 1, m_dLambdas[3] = {0, 0, 0}; Bool m_bEncodeDQP = false;
  uint64_t m_uiPicTotalBits = 0, m_uiPicDist = 0; Double m_dPicRdCost = 0;
};

// TComLoopFilter::loopFilterPic (TComLoopFilter.cpp:130): deblocks the picture compressSlice just left in the device slot
class TComLoopFilter {
public:
  Void init(TEncTop *pcEncTop) { m_pcEncTop = pcEncTop; }
  Void loopFilterPic(TComPic *pcPic);
private:
  TEncTop *m_pcEncTop = nullptr;
};
// TEncSampleAdaptiveOffset::SAOProcess (TEncSampleAdaptiveOffset.cpp:259); m_saoDisabledRate lives here as in the reference
class TEncSampleAdaptiveOffset {
public:
  Void init(TEncTop *pcEncTop) { m_pcEncTop = pcEncTop; }
  Void SAOProcess(TComPic *pPic, Bool *sliceEnabled, const Double *lambdas);
private:
  TEncTop *m_pcEncTop = nullptr; Double m_saoDisabledRate[3][8] = {};
};

// TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64): the quadrant sums come from the device (hm355_preanalyze), the double arithmetic is the reference's
class TEncPreanalyzer {
public:
  Void init(TEncTop *pcEncTop) { m_pcEncTop = pcEncTop; }
  Void xPreanalyze(TComPic *pcPic);
private:
  TEncTop *m_pcEncTop = nullptr;
};

class TEncGOP {
public:
  Void init(TEncTop *pcTEncTop);
  // TEncGOP::compressGOP (TEncGOP.cpp:527): one picture per call for the all-intra GOP (GOPSize 1)
  Void compressGOP(Int iPOCLast, Int iNumPicRcvd, std::list<TComPic *> &rcListPic);
  const std::vector<TComPic *> &getCodedPictures() const { return m_codedPics; }   // every picture encoded so far, in coding order
private:
  std::vector<TComPic *> m_codedPics;
  // reference picture set and lists of a P / B slice (TEncTop::selectReferencePictureSet, the extra sets TAppEncCfg::xCheckParameter builds for
  // the start of the sequence, TComSlice::setRefPicList): no list modification, no long-term pictures
  Void xSetReferences(TComSlice *pcSlice, Int pocCurr, Int iGOPid, std::list<TComPic *> &rcListPic);
  TEncTop *m_pcEncTop = nullptr; TEncSlice *m_pcSliceEncoder = nullptr; TComLoopFilter *m_pcLoopFilter = nullptr; TEncSampleAdaptiveOffset *m_pcSAO = nullptr;
};

class TEncTop : public TEncCfg {
public:
  Void create();                     // TEncTop.cpp:87: allocates the device context
  Void destroy();
  Void init();                       // TEncTop.cpp:187
  // TEncTop::encode (TEncTop.cpp:259): takes one original picture, encodes when a GOP is complete
  Void encode(Bool flush, TComPicYuv *pcPicYuvOrg, std::list<TComPic *> &rcListPicOut, Int &iNumEncoded);
  TEncSlice *getSliceEncoder() { return &m_cSliceEncoder; }
  TComLoopFilter *getLoopFilter() { return &m_cLoopFilter; }
  TEncSampleAdaptiveOffset *getSAO() { return &m_cEncSAO; }
  hm355_ctx *getDeviceContext() { return m_ctx; }
  TEncGOP *getGOPEncoder() { return &m_cGOPEncoder; }
  Void setEncCABACTableIdx(Int i) { m_encCABACTableIdx = i; } Int getEncCABACTableIdx() const { return m_encCABACTableIdx; }   // TComPPS::m_encCABACTableIdx
private:
  Int m_encCABACTableIdx = I_SLICE;
  TEncPreanalyzer m_cPreanalyzer;
  hm355_ctx *m_ctx = nullptr; TEncGOP m_cGOPEncoder; TEncSlice m_cSliceEncoder; TComLoopFilter m_cLoopFilter; TEncSampleAdaptiveOffset m_cEncSAO;
  std::list<TComPic *> m_cListPic; Int m_iPOCLast = -1, m_iNumPicRcvd = 0;
};
