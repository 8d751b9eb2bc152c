// hm355 device/host shared data layout (HBM-resident structures of the CTU RD search).
// No reference code is copied here: the layouts are our own; where a field mirrors an array of the
// reference's TComDataCU the comment names it (source/Lib/TLibCommon/TComDataCU.h:86-157).
#pragma once
#include <stdint.h>

typedef int16_t Pel;      // TypeDef.h:692  (RExt__HIGH_BIT_DEPTH_SUPPORT=0)
typedef int32_t TCoeff;   // TypeDef.h:693

// ---- CABAC estimator state: one byte per context model + the Q15 fractional-bit accumulator ----
// context numbering is ours; group sizes follow ContextTables.h:51-161 (intra subset)
enum {
  C_SPLIT = 0, C_PART = 3, C_INTRA_LUMA = 7, C_CHROMA_PRED = 8, C_SUBDIV = 10, C_QT_CBF = 13, C_SIG_CG = 23,
  C_SIG = 27, C_LASTX = 71, C_LASTY = 101, C_ONE = 131, C_ABS = 155, C_TSKIP = 161,
  /* inter syntax */ C_SKIP = 163, C_MRG_FLAG = 166, C_MRG_IDX = 167, C_PRED_MODE = 168, C_INTER_DIR = 169, C_MVD = 174, C_REF = 176,
  C_ROOT_CBF = 178, C_MVP_IDX = 179, /* cu_qp_delta_abs */ C_DQP = 180, HM_NUM_CTX = 183
};
struct Cabac {
  uint8_t s[184];          // HM_NUM_CTX used, padded to 8-byte multiple
  uint64_t frac;           // TEncBinCABAC::m_fracBits
};
enum { CI_CURR_BEST = 0, CI_NEXT_BEST, CI_TEMP_BEST, CI_QT_TRAFO_TEST, CI_QT_TRAFO_ROOT, CI_NUM };   // TypeDef.h:477-486 minus the unused CI_CHROMA_INTRA

// ---- per-CTU decision arrays, 256 4x4 partitions in z-scan order ----
struct CtuMeta {
  uint8_t depth[256], part[256], pred[256], dirL[256], dirC[256], tr[256], cbf[3][256], ts[3][256];
};
struct CtuStat { double cost; uint32_t bits, dist; };
// ---- inter (P / B slice) per-CTU arrays: TComDataCU m_skipFlag, m_pbMergeFlag, m_puhMergeIndex, m_puhInterDir, m_acCUMvField[2], m_apiMVPIdx/Num ----
struct MvD { int16_t x, y; };
struct InterMeta {
  uint8_t skip[256], mrg[256], mrgIdx[256], interDir[256];
  MvD mv[2][256], mvd[2][256];
  int8_t refIdx[2][256], mvpIdx[2][256], mvpNum[2][256];
};
#define HM_REF_MARGIN 80              // TComPicYuv margin: max CU width + 16
// one reference picture as the decoded picture buffer holds it
struct RefPicDev {
  const Pel *plane[3];                // border-extended planes, pointing at sample (0,0)
  int32_t stride[3];
  int32_t poc, isLongTerm;
  const uint8_t *predMode;            // motion field after TComPic::compressMotion: [numCtus*256]
  const MvD *mv[2]; const int8_t *refIdx[2];
  int32_t refPoc[2][16], refLT[2][16];
};
// slice-level inter parameters of one picture slot
struct InterPic {
  int32_t sliceType, poc, numRefIdx[2];
  RefPicDev ref[2][16];
  int32_t colFromL0, colRefIdx, tmvp, mvdL1Zero, maxMergeCand, checkLDC, cabacInitType;
  uint32_t lambdaMotionSAD, lambdaMotionSSE;
  MvD integerMv2Nx2N[2][16];          // TEncSearch::m_integerMv2Nx2N on entry to the slice
  int32_t list1ToList0[16];           // TComSlice::m_list1IdxToList0Idx: list-1 entry -> list-0 entry holding the same picture, or -1
};

// planes of one CTU-sized scratch picture: Y 64x64 at 0, Cb 32x32 at 4096, Cr 32x32 at 5120
#define HM_PLANE_OFF(c) ((c) == 0 ? 0 : ((c) == 1 ? 4096 : 5120))
#define HM_PLANE_STRIDE(c) ((c) == 0 ? 64 : 32)
// coefficients of one CTU in the reference's packing: Y at 0 (z*16), Cb at 4096 (z*4), Cr at 5120 (z*4)
#define HM_COEF_CTU 6144

struct Best {              // best mode of one CU depth (the role of m_ppcBestCU[d] / m_ppcRecoYuvBest[d])
  CtuMeta m;
  InterMeta im;
  TCoeff coef[HM_COEF_CTU];
  Pel reco[HM_COEF_CTU];
};

// per-workgroup scratch in HBM (one CTU in flight per workgroup)
struct WorkSpace {
  Best best[4];
  Pel pred[HM_COEF_CTU], resi[HM_COEF_CTU], reco[HM_COEF_CTU];
  Pel qtRec[4][HM_COEF_CTU];         // m_pcQTTempTComYuv[layer]
  TCoeff qtCoef[4][HM_COEF_CTU];     // m_ppcQTTempCoeff[comp][layer]
  MvD intMv[2][16];                  // TEncSearch::m_integerMv2Nx2N[list][refIdx] of the search in progress (read and written once per motion search)
  Pel tsPred[3][16], tsRec[3][16]; TCoeff tsCoef[3][16];   // the parked first trial of a 4x4 transform-skip decision (wave-uniform path; rare since the candidates-in-lanes paths)
  double costCoeff[1024];            // RDOQ of 32x32 blocks: cost of the positions that keep a non-zero level (everything else of its per-position state is in LDS)
  Cabac slot[4 * CI_NUM + 3];        // m_pppcRDSbacCoder[depth][CI_*] snapshots (the live coder stays in LDS); depth 4 only holds TEMP_BEST/QT_TRAFO_*
  Pel tmpPred[HM_COEF_CTU];          // m_tmpYuvPred (merge / ME prediction error)
  Pel resiBest[HM_COEF_CTU];         // m_ppcResiYuvBest[depth]
  Pel mcTmp[72 * 64];                // first interpolation stage of a block (m_filteredBlockTmp)
  Pel mcBlk[64 * 64];                // interpolated candidate block of the fractional search
  Pel yuvPred[2][HM_COEF_CTU];       // m_acYuvPred[list]: bi-prediction halves (14-bit) / the other list's prediction during the bi search
  Pel orgBi[64 * 64];                // m_cYuvPredTemp: 2*org - other prediction (luma)
  // per-(list, refIdx) state of TEncSearch::predInterSearch (:3098-3135) for the PU under search
  struct PuSearch {
    MvD mvTemp[2][16], mvPred[2][16], mvPredBi[2][16], cand[2][16][2];
    int8_t mvpIdx[2][16], mvpIdxBi[2][16], mvpNum[2][16];
    uint32_t costTempL0[16], bitsTempL0[16];
  } ps;
  struct { MvD mv[5][2]; int32_t ref[5][2]; uint8_t dir[5]; int32_t num; } mrg2N;   // merge list of xCheckRDCostMerge2Nx2N (outlives the per-PU list in LDS)
  uint8_t tmpTr[256], tmpCbf[3][256], tmpTs[3][256], saveCbf[3][256], saveTs[3][256];
  struct { int32_t ctuQp, refQp, flag, pad; } dq;   // cu_qp_delta: QP of the CTU under search, its predictor, TEncCu::m_bEncodeDQP
  TCoeff teamCoef[HM_COEF_CTU];      // a team helper's private coefficient area (the team's main wavefront works in the picture's own, hm355_team.h)
  InterMeta teamIm;                  // ... and its private copy of the CTU's motion arrays (P / B slices)
  MvD teamTok[3][2][16];             // main wavefront of a team: m_integerMv2Nx2N after the unsplit CU of depth 0..2, where its sub-CUs start from
};

// ---- cu_qp_delta (adaptive QP / rate control; MaxCuDQPDepth 0, i.e. one quantisation group per CTU) ----
struct QpTab { int32_t qpPer[2], qpRem[2]; double errScale[2][4]; int64_t rdFactor[2]; };   // what hm355_fill_slice_params derives from one QP
struct CtuDqp {                      // what a searched CTU leaves for the next CTU, the loop filter and the bitstream pass
  int8_t qp, refQp, lastQp;          // its QP, the predictor (TComDataCU::getRefQP), the QP of its last coded CU (getLastCodedQP of the next CTU)
  uint8_t flagOut;                   // TEncCu::m_bEncodeDQP after its encodeCtu
  int16_t firstZ, pad;               // first CU (z order) with a coded block, 256 if none: partitions before it carry refQp, the rest qp (m_phQP)
};
struct DqpPic {
  int32_t flagIn, sliceQp;           // m_bEncodeDQP on entry to the slice
  const int8_t *ctuQp;               // [numCtus] QP of every CTU (TEncCu::xComputeQP / TEncRateCtrl::getRCQP)
  CtuDqp *out;                       // [numCtus]
  const uint8_t *rowFlag;            // [hCtu] WaveFrontSynchro: m_bEncodeDQP assumed at the first CTU of each row (checked by the host afterwards)
  QpTab tab[64];                     // indexed by QP + 12
};

// lookup tables generated on the host at create time (scan orders: TComRom.cpp:52-225)
struct Tables {
  uint8_t z2r[256], r2z[256];
  uint16_t scan[3][4][1024];         // [scanType][log2-2][pos] grouped-4x4 scan -> raster position
  uint16_t scanCG[3][4][64];         // [scanType][log2-2][cg]  scan of the coefficient-group grid
};

// one picture resident in HBM
struct FrameBuf {
  Pel *org[3], *rec[3];              // planes padded to whole CTUs (stride = wCtu*64 / wCtu*32)
  CtuMeta *meta;                     // [numCtus]
  TCoeff *coef;                      // [numCtus][HM_COEF_CTU]
  CtuStat *stat;                     // [numCtus]
  Cabac *endState;                   // [numCtus] estimator state after encodeCtu of that CTU
  uint32_t *done;                    // [numCtus] == run epoch once the CTU's results are published (persistent scheduler)
  InterMeta *imeta;                  // [numCtus] (P / B slices; NULL for I slices)
  InterPic *ip;                      // slice-level inter parameters (NULL for I slices)
  MvD *intMv;                        // [numCtus][2][16] m_integerMv2Nx2N as each CTU left it (carried in coding order)
  DqpPic *dqp;                       // cu_qp_delta state of the picture (NULL: disabled, every CU at the slice QP)
  // slice parameters (TEncSlice::setUpLambda, TEncSlice.cpp:132-159)
  double lambda, sqrtLambda, lambdaC, chromaWeight;
  double errScale[2][4];             // [luma/chroma][log2-2]  TComTrQuant::setErrScaleCoeff :2933
  int64_t rdFactor[2];               // sign-bit-hiding factor  TComTrQuant.cpp:2382-2386
  int32_t qp, qpPer[2], qpRem[2];
};

struct Params {
  int32_t width, height, bitDepth, wpp;
  int32_t wCtu, hCtu, stride[3];
  const Tables *tab;
  WorkSpace *ws;
  FrameBuf *frames;
  unsigned long long *prof;          // diagnostic builds only (HM355_PROFILE), else NULL
  int32_t fewWaves;                  // the launch cannot fill the device: prefer the shortest dependency chain over the fewest instructions
  Pel *teamWin;                      // team launches: the helpers' private reconstruction windows, HM_TEAM_HELPERS per team (hm355_team.h)
  uint64_t teamWinStride;            // samples per helper window
};

struct WorkItem { int32_t frame, ctuX, ctuY, pad; };
