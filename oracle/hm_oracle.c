/* TEST INFRASTRUCTURE ONLY -- see hm_oracle.h.
 *
 * Scalar C restatement of the I-slice hot path of liron88/HM-16.2 (all paths relative to
 * /root/reference/source/Lib).  Written from the reference's behaviour, in its own layout:
 * one frame-level store (per-CTU z-scan metadata + HM-packed coefficients + planar recon),
 * trial modes are evaluated in place and the best mode of each depth is kept as a snapshot.
 * Every function names the reference function (file:line) it restates.
 */
#include "hm_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

typedef int16_t Pel;
typedef int32_t TCoeff;

#define MAX_DOUBLE 1.7e+308 /* TLibCommon/CommonDef.h */
#define PLANAR_IDX 0
#define DC_IDX 1
#define HOR_IDX 10
#define VER_IDX 26
#define DM_CHROMA_IDX 36
#define SIZE_2Nx2N 0
#define SIZE_2NxN 1
#define SIZE_Nx2N 2
#define SIZE_NxN 3
#define SIZE_2NxnU 4
#define SIZE_2NxnD 5
#define SIZE_nLx2N 6
#define SIZE_nRx2N 7
#define SIZE_NONE 8  /* NUMBER_OF_PART_SIZES */
#define MODE_INTER 0
#define MODE_INTRA 1
#define MODE_NONE 2  /* NUMBER_OF_PREDICTION_MODES */
#define B_SLICE 0
#define P_SLICE 1
#define I_SLICE 2
#define SCAN_DIAG 0
#define SCAN_HOR 1
#define SCAN_VER 2

static FILE *g_trace;
void hmo_set_trace(const char *path) { if (g_trace) fclose(g_trace); g_trace = path ? fopen(path, "w") : NULL; }

/* ============================================================================================ */
/* tables                                                                                        */
/* ============================================================================================ */
static int g_init_done;
static int Z2R[256], R2Z[256];           /* TComRom.cpp:256-290 initZscanToRaster/initRasterToZscan */
static uint16_t *SCAN[3][4];              /* grouped-4x4 scan, [type][log2-2]   TComRom.cpp:140-225 */
static uint16_t *SCANCG[3][4];            /* ungrouped scan of the CG grid, [type][log2(cgw)]       */
static int T32[32][32];                   /* TComRom.cpp:456-484 g_aiT32 (4/8/16 are sub-sampled rows) */
static const int DST4[4][4] = { {29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29} };
static const int QUANT_SCALES[6] = {26214, 23302, 20560, 18396, 16384, 14564};     /* TComRom.cpp:321 */
static const int INV_QUANT_SCALES[6] = {40, 45, 51, 57, 64, 72};                   /* TComRom.cpp:326 */
static const uint8_t CHROMA_SCALE_420[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51 };
static const uint8_t GROUP_IDX[32] = {0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9};   /* TComRom.cpp g_uiGroupIdx */
static const uint8_t MIN_IN_GROUP[10] = {0,1,2,3,4,6,8,12,16,24};
static const uint8_t CTX_IND_MAP_4x4[16] = {0,1,4,5, 2,3,4,5, 6,6,8,8, 7,7,8,8};
static const uint8_t INTRA_MODE_NUM_FAST[6] = {3, 8, 8, 3, 3, 3};                /* TComRom.cpp:513 (index = log2-1) */
static const uint8_t INTRA_FILTER[5] = {10, 7, 1, 0, 10};                         /* TComPrediction.cpp:49 */

/* CABAC tables: ContextModel.cpp:66-128 (FAST_BIT_EST variant) */
static const uint8_t NEXT_MPS[128] = {
  2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,
  34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 97,
  98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 124, 125, 126, 127 };
static const uint8_t NEXT_LPS[128] = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
static const int32_t ENTROPY_BITS[128] = {
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };

/* context layout (own numbering); sizes follow ContextTables.h:51-161 */
enum {
  C_SPLIT = 0, C_PART = 3, C_INTRA_LUMA = 7, C_CHROMA_PRED = 8, C_SUBDIV = 10, C_QT_CBF = 13, C_SIG_CG = 23,
  C_SIG = 27, C_LASTX = 71, C_LASTY = 101, C_ONE = 131, C_ABS = 155, C_TSKIP = 161,
  /* inter syntax */ C_SKIP = 163, C_MRG_FLAG = 166, C_MRG_IDX = 167, C_PRED_MODE = 168, C_INTER_DIR = 169, C_MVD = 174, C_REF = 176,
  C_ROOT_CBF = 178, C_MVP_IDX = 179, /* cu_qp_delta_abs */ C_DQP = 180, NUM_CTX = 183
};
/* I-slice initialisation values, ContextTables.h:170-502 (row [2] of each table; CNU = 154) */
static const uint8_t CTX_INIT_I[NUM_CTX] = {
  /* split */ 139, 141, 157,
  /* part size */ 184, 154, 154, 154,
  /* intra luma */ 184,
  /* chroma pred */ 63, 139,
  /* trans subdiv */ 153, 138, 138,
  /* qt cbf */ 111, 141, 154, 154, 154,   94, 138, 182, 154, 154,
  /* sig cg */ 91, 171, 134, 141,
  /* sig luma 28 */ 111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 141,
  /* sig chroma 16 */ 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 111,
  /* last x: luma 15, chroma 15 */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,  108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* last y */ 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79,  108, 123, 63, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* one: luma 16, chroma 8 */ 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152,  140, 179, 166, 182, 140, 227, 122, 197,
  /* abs: luma 4, chroma 2 */ 138, 153, 136, 167, 152, 152,
  /* transform skip */ 139, 139,
  /* skip */ 154, 154, 154, /* merge flag, idx */ 154, 154, /* pred mode */ 154, /* inter dir */ 154, 154, 154, 154, 154, /* mvd */ 154, 154,
  /* ref idx */ 154, 154, /* root cbf */ 154, /* mvp idx */ 154, /* delta qp */ 154, 154, 154
};
/* P-slice initialisation values (row [1] of each table) */
static const uint8_t CTX_INIT_P[NUM_CTX] = {
  /* split */ 107, 139, 126,
  /* part size */ 154, 139, 154, 154,
  /* intra luma */ 154,
  /* chroma pred */ 152, 139,
  /* trans subdiv */ 124, 138, 94,
  /* qt cbf */ 153, 111, 154, 154, 154,   149, 107, 167, 154, 154,
  /* sig cg */ 121, 140, 61, 154,
  /* sig luma 28 */ 155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 140,
  /* sig chroma 16 */ 170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140, 140,
  /* last x */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94,  108, 123, 108, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* last y */ 125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94,  108, 123, 108, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* one: luma 16, chroma 8 */ 154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137,  169, 194, 166, 167, 154, 167, 137, 182,
  /* abs: luma 4, chroma 2 */ 107, 167, 91, 122, 107, 167,
  /* transform skip */ 139, 139,
  /* skip */ 197, 185, 201, /* merge flag, idx */ 110, 122, /* pred mode */ 149, /* inter dir */ 95, 79, 63, 31, 31, /* mvd */ 140, 198,
  /* ref idx */ 153, 153, /* root cbf */ 79, /* mvp idx */ 168, /* delta qp */ 154, 154, 154
};
/* B-slice initialisation values (row [0] of each table) */
static const uint8_t CTX_INIT_B[NUM_CTX] = {
  /* split */ 107, 139, 126,
  /* part size */ 154, 139, 154, 154,
  /* intra luma */ 183,
  /* chroma pred */ 152, 139,
  /* trans subdiv */ 224, 167, 122,
  /* qt cbf */ 153, 111, 154, 154, 154,   149, 92, 167, 154, 154,
  /* sig cg */ 121, 140, 61, 154,
  /* sig luma 28 */ 170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 140,
  /* sig chroma 16 */ 170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140, 140,
  /* last x */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79,  108, 123, 93, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* last y */ 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79,  108, 123, 93, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
  /* one: luma 16, chroma 8 */ 154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122,  169, 208, 166, 167, 154, 152, 167, 182,
  /* abs: luma 4, chroma 2 */ 107, 167, 91, 107, 107, 167,
  /* transform skip */ 139, 139,
  /* skip */ 197, 185, 201, /* merge flag, idx */ 154, 137, /* pred mode */ 134, /* inter dir */ 95, 79, 63, 31, 31, /* mvd */ 169, 198,
  /* ref idx */ 153, 153, /* root cbf */ 79, /* mvp idx */ 168, /* delta qp */ 154, 154, 154
};

static void gen_scan(int w, int h, int stride, int type, int offx, int offy, uint16_t *out, int count)
{ /* ScanGenerator, TComRom.cpp:52-137 */
  int line = 0, col = 0;
  for (int i = 0; i < count; i++) {
    out[i] = (uint16_t)((line + offy) * stride + col + offx);
    if (type == SCAN_DIAG) {
      if (col == w - 1 || line == 0) { line += col + 1; col = 0; if (line >= h) { col += line - (h - 1); line = h - 1; } }
      else { col++; line--; }
    } else if (type == SCAN_HOR) { if (col == w - 1) { line++; col = 0; } else col++; }
    else { if (line == h - 1) { col++; line = 0; } else line++; }
  }
}

static void z2r_rec(int maxd, int d, int start, int **cur)
{
  int stride = 1 << (maxd - 1);
  if (d == maxd) { **cur = start; (*cur)++; return; }
  int step = stride >> d;
  z2r_rec(maxd, d + 1, start, cur); z2r_rec(maxd, d + 1, start + step, cur);
  z2r_rec(maxd, d + 1, start + step * stride, cur); z2r_rec(maxd, d + 1, start + step * stride + step, cur);
}

static void init_tables(void)
{
  if (g_init_done) return;
  int *p = Z2R; z2r_rec(5, 1, 0, &p);
  for (int i = 0; i < 256; i++) R2Z[Z2R[i]] = i;
  for (int t = 0; t < 3; t++)
    for (int l = 0; l < 4; l++) {
      int n = 4 << l, g = n >> 2;
      SCAN[t][l] = (uint16_t *)malloc(sizeof(uint16_t) * n * n);
      SCANCG[t][l] = (uint16_t *)malloc(sizeof(uint16_t) * g * g);
      gen_scan(g, g, g, t, 0, 0, SCANCG[t][l], g * g);
      for (int gi = 0; gi < g * g; gi++) {
        int gx = SCANCG[t][l][gi] % g, gy = SCANCG[t][l][gi] / g;
        gen_scan(4, 4, n, t, gx * 4, gy * 4, SCAN[t][l] + gi * 16, 16);
      }
    }
  /* HEVC core transform: first column of the 32-point matrix, the rest follows the cosine index law */
  static const int c[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
  for (int k = 0; k < 32; k++)
    for (int n = 0; n < 32; n++) {
      int m = (k * (2 * n + 1)) % 128, v;
      if (m <= 32) v = c[m]; else if (m <= 64) v = -c[64 - m]; else if (m <= 96) v = -c[m - 64]; else v = c[128 - m];
      T32[k][n] = v;
    }
  g_init_done = 1;
}

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* ============================================================================================ */
/* CABAC bit estimator (TEncBinCABACCounter, TEncBinCoderCABACCounter.cpp:56-131)                */
/* ============================================================================================ */
/* The bitstream pass (hm_oracle_bits.inc) drives the same syntax functions with the arithmetic coder attached (be != NULL):
   every bin then also goes through TEncBinCABAC (TEncBinCoderCABAC.cpp:69-437). */
struct BinEnc;
typedef struct { uint8_t s[NUM_CTX]; uint8_t pad[4]; uint64_t frac; struct BinEnc *be; } Cabac;
static void be_bin(struct BinEnc *b, int ctx, int st, int bin);
static void be_ep(struct BinEnc *b, uint32_t val, int n);
static void be_trm(struct BinEnc *b, int bin);

static int g_slice_type = I_SLICE;        /* selects the initialisation table (TEncSbac::resetEntropy uses the slice type) */
static void cabac_init(Cabac *c, int qp)
{ /* ContextModel::init, ContextModel.cpp:55-64; TEncSbac::resetEntropy, TEncSbac.cpp:106-161 */
  const uint8_t *tab = g_slice_type == I_SLICE ? CTX_INIT_I : (g_slice_type == P_SLICE ? CTX_INIT_P : CTX_INIT_B);
  qp = clip3(0, 51, qp);
  for (int i = 0; i < NUM_CTX; i++) {
    int iv = tab[i];
    int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
    int mps = st >= 64;
    c->s[i] = (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
  }
  c->frac = 0; c->be = NULL;
}
static inline void enc_bin(Cabac *c, int ctx, int bin)
{
  uint8_t st = c->s[ctx];
  if (c->be) be_bin(c->be, ctx, st, bin);
  c->frac += (uint64_t)ENTROPY_BITS[st ^ bin];
  c->s[ctx] = ((st & 1) == bin) ? NEXT_MPS[st] : NEXT_LPS[st];
}
static inline void enc_ep(Cabac *c, int n) { if (c->be) abort(); c->frac += (uint64_t)32768 * (uint64_t)n; }   /* estimator-only call sites */
static inline void enc_epv(Cabac *c, uint32_t val, int n) { if (c->be) be_ep(c->be, val, n); c->frac += (uint64_t)32768 * (uint64_t)n; }  /* n bypass bins, first bin = MSB of val */
static inline void enc_trm(Cabac *c, int bin) { if (c->be) be_trm(c->be, bin); c->frac += (uint64_t)ENTROPY_BITS[126 ^ bin]; }
static inline void reset_bits(Cabac *c) { c->frac &= 32767; }          /* TEncBinCABAC::resetBits, TEncBinCoderCABAC.cpp:161 */
static inline uint32_t num_bits(const Cabac *c) { return (uint32_t)(c->frac >> 15); }
static inline int ebits(const Cabac *c, int ctx, int bin) { return ENTROPY_BITS[c->s[ctx] ^ bin]; }

/* ============================================================================================ */
/* encoder state                                                                                 */
/* ============================================================================================ */
enum { CI_CURR_BEST = 0, CI_NEXT_BEST, CI_TEMP_BEST, CI_CHROMA_INTRA, CI_QT_TRAFO_TEST, CI_QT_TRAFO_ROOT, CI_NUM };

typedef struct { int16_t x, y; } Mv;
typedef struct {
  uint8_t depth[256], part[256], pred[256], dirL[256], dirC[256], tr[256], cbf[3][256], ts[3][256];
  /* inter (TComDataCU m_skipFlag, m_pbMergeFlag, m_puhMergeIndex, m_puhInterDir, m_acCUMvField[2], m_apiMVPIdx/Num) */
  uint8_t skip[256], mrg[256], mrgIdx[256], interDir[256];
  Mv mv[2][256], mvd[2][256];
  int8_t refIdx[2][256], mvpIdx[2][256], mvpNum[2][256];
  int8_t qp[256];                  /* m_phQP */
} CtuMeta;

typedef struct {
  /* snapshot of the best mode of one depth (the reference's m_ppcBestCU[d] + m_ppcRecoYuvBest[d]) */
  CtuMeta m;                       /* only [cuZ, cuZ+parts) used */
  TCoeff coef[3][4096];            /* CTU-relative HM packing, only the CU range used */
  Pel reco[3][64 * 64];            /* CTU-relative planes (chroma uses stride 32) */
  double cost; uint32_t bits, dist;
} Best;

#define REF_MARGIN 80                      /* TComPicYuv margin: g_uiMaxCUWidth + 16 (TComPic::create) */
typedef struct {
  int poc, isLongTerm, sliceType;
  Pel *buf[3]; Pel *plane[3]; int stride[3];      /* border-extended planes, plane[] points at sample (0,0) */
  /* motion field as later pictures see it (after TComPic::compressMotion): per CTU, 256 partitions */
  const uint8_t *predMode; const int16_t *mv[2]; const int8_t *refIdx[2];
  int numRef[2], refPoc[2][16], refLT[2][16];
} RefPic;

typedef struct {
  int sliceType, poc;
  int numRefIdx[2];
  RefPic *ref[2][16];
  int colFromL0, colRefIdx, tmvp, mvdL1Zero, maxMergeCand, checkLDC;
  uint32_t lambdaMotionSAD, lambdaMotionSSE;        /* TComRdCost::m_uiLambdaMotionSAD/SSE[0] */
  /* TComRdCost motion-cost state (getMotionCost / setPredictor / setCostScale) */
  uint32_t mcost; Mv mvPredictor; int costScale;
  /* TEncSearch::m_integerMv2Nx2N[list][refIdx]: integer MV of the last 2Nx2N ME on that reference (persists across CUs) */
  Mv integerMv2Nx2N[2][16];
  int list1ToList0[16];            /* TComSlice::m_list1IdxToList0Idx (setList1IdxToList0Idx, TComSlice.cpp) */
} InterSlice;

typedef struct {
  hmo_cfg cfg;
  InterSlice *is;                  /* NULL for I slices */
  int wCtu, hCtu;
  int stride[3], ph[3];            /* plane strides / heights padded to whole CTUs */
  Pel *org[3], *rec[3];
  CtuMeta *meta;                   /* per CTU */
  TCoeff *coef[3];                 /* per CTU: 4096 / 1024 / 1024 */
  double lambda, sqrtLambda, lambdaC, chromaWeight;
  int qpRem[3], qpPer[3];
  /* cu_qp_delta (MaxCuDQPDepth 0: the CTU is the quantisation group) */
  const hmo_dqp *dq;               /* NULL: cu_qp_delta disabled, every CU at the slice QP */
  int dqpFlag;                     /* TEncCu::m_bEncodeDQP */
  int ctuQp, refQp;                /* QP of the CTU under search (xComputeQP / rate control) and its predictor (TComDataCU::getRefQP) */
  int8_t *lastQp;                  /* per CTU: QP of its last coded CU (TComDataCU::getLastCodedQP of the next CTU) */
  /* current CTU */
  int ctuX, ctuY, ctuAddr;
  CtuMeta *cm; TCoeff *cc[3];
  Cabac cur;                       /* m_pcRDGoOnSbacCoder */
  Cabac slot[5][CI_NUM];           /* m_pppcRDSbacCoder[depth][CI_*], TEncTop.cpp:120-146 */
  Cabac wppSync;                   /* m_entropyCodingSyncContextState */
  Best best[4];
  /* per-trial scratch, CTU-relative */
  Pel pred[3][64 * 64], resi[3][64 * 64], reco[3][64 * 64];
  Pel tmpPred[3][64 * 64];         /* m_tmpYuvPred (merge / ME prediction error) */
  Pel yuvPred[2][3][64 * 64];      /* m_acYuvPred[list] (bi-prediction halves / the other list's prediction during the bi search) */
  Pel orgBi[64 * 64];              /* m_cYuvPredTemp: 2*org - other prediction (luma), CTU-relative */
  Pel resiBest[3][64 * 64];        /* m_ppcResiYuvBest[depth] */
  Pel qtRec[4][3][64 * 64];        /* m_pcQTTempTComYuv[layer] */
  TCoeff qtCoef[3][4][4096];       /* m_ppcQTTempCoeff[comp][layer] */
  TCoeff tsCoef[3][1024];          /* m_pcQTTempTUCoeff */
  Pel tsRec[3][32 * 32];           /* m_pcQTTempTransformSkipTComYuv (block-local) */
  Pel tsPred[3][32 * 32];          /* m_pSharedPredTransformSkip */
  uint8_t tmpTr[256], tmpCbf[3][256], tmpTs[3][256]; /* m_puhQTTempTrIdx/Cbf/TransformSkipFlag */
  /* reference sample lines of the current block: [filtered?][0]=top-left, then 2N samples */
  Pel refTop[2][129], refLeft[2][129];
} Enc;

static inline double calc_rd_cost(const Enc *e, uint32_t bits, uint32_t dist)
{ /* TComRdCost::calcRdCost, TComRdCost.cpp:56-123 (DF_DEFAULT, COST_STANDARD_LOSSY) */
  double c = floor((double)dist + ((double)bits * e->lambda) + 0.5);
  if (g_trace) fprintf(g_trace, "RD %u %u %.1f\n", bits, dist, c);
  return c;
}

/* ============================================================================================ */
/* distortion primitives (TComRdCost.cpp)                                                        */
/* ============================================================================================ */
uint32_t hmo_sad(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int sub_shift, int bit_depth)
{ /* xGetSAD*, TComRdCost.cpp:465-962 */
  uint32_t sum = 0; int step = 1 << sub_shift;
  for (int y = 0; y < h; y += step) for (int x = 0; x < w; x++) sum += (uint32_t)abs(org[y * so + x] - cur[y * sc + x]);
  sum <<= sub_shift;
  return sum >> (bit_depth - 8);
}
uint32_t hmo_sse(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int bit_depth)
{ /* xGetSSE*, TComRdCost.cpp:970-1318 */
  uint32_t sum = 0; int shift = (bit_depth - 8) << 1;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { int d = org[y * so + x] - cur[y * sc + x]; sum += (uint32_t)((d * d) >> shift); }
  return sum;
}
static uint32_t had_block(const int16_t *org, int so, const int16_t *cur, int sc, int n)
{ /* xCalcHADs4x4 / xCalcHADs8x8, TComRdCost.cpp:1343,1439: sum |H D H| with the final rounding shift */
  int d[64], t[64];
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) d[y * n + x] = org[y * so + x] - cur[y * sc + x];
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < n; i++) {       /* transform each row in place, then transpose */
      int *r = d + i * n;
      for (int len = 1; len < n; len <<= 1)
        for (int b = 0; b < n; b += len << 1)
          for (int k = 0; k < len; k++) { int a0 = r[b + k], a1 = r[b + k + len]; r[b + k] = a0 + a1; r[b + k + len] = a0 - a1; }
    }
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) t[x * n + y] = d[y * n + x];
    memcpy(d, t, sizeof(int) * n * n);
  }
  uint32_t s = 0;
  for (int i = 0; i < n * n; i++) s += (uint32_t)abs(d[i]);
  return n == 8 ? ((s + 2) >> 2) : ((s + 1) >> 1);
}
uint32_t hmo_hads(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int bit_depth)
{ /* xGetHADs, TComRdCost.cpp:1537-1606 */
  uint32_t sum = 0; int n = ((w % 8 == 0) && (h % 8 == 0)) ? 8 : 4;
  for (int y = 0; y < h; y += n) for (int x = 0; x < w; x += n) sum += had_block(org + y * so + x, so, cur + y * sc + x, sc, n);
  return sum >> (bit_depth - 8);
}

/* ============================================================================================ */
/* transforms (TComTrQuant.cpp:387-935)                                                          */
/* ============================================================================================ */
static inline int tmat(int n, int use_dst, int k, int j) { return use_dst ? DST4[k][j] : T32[k * (32 / n)][j]; }

void hmo_fwd_transform(int bit_depth, const int32_t *block, int32_t *coeff, int n, int use_dst)
{ /* xTrMxN, TComTrQuant.cpp:836-890: two partialButterfly passes == two integer matrix products */
  int l2 = ilog2(n), s1 = l2 + bit_depth + 6 - 15, s2 = l2 + 6;
  int32_t tmp[32 * 32];
  int a1 = s1 > 0 ? 1 << (s1 - 1) : 0, a2 = 1 << (s2 - 1);
  for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
    int32_t acc = 0; for (int i = 0; i < n; i++) acc += tmat(n, use_dst, k, i) * block[j * n + i];
    tmp[k * n + j] = (acc + a1) >> s1;
  }
  for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {
    int32_t acc = 0; for (int i = 0; i < n; i++) acc += tmat(n, use_dst, k, i) * tmp[j * n + i];
    coeff[k * n + j] = (acc + a2) >> s2;
  }
}
void hmo_inv_transform(int bit_depth, const int32_t *coeff, int32_t *block, int n, int use_dst)
{ /* xITrMxN, TComTrQuant.cpp:894-935 */
  int s1 = 7, s2 = 20 - bit_depth;
  int32_t tmp[32 * 32];
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
    int32_t acc = 0; for (int k = 0; k < n; k++) acc += tmat(n, use_dst, k, i) * coeff[k * n + j];
    tmp[j * n + i] = clip3(-32768, 32767, (acc + (1 << (s1 - 1))) >> s1);
  }
  for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
    int32_t acc = 0; for (int k = 0; k < n; k++) acc += tmat(n, use_dst, k, i) * tmp[k * n + j];
    block[j * n + i] = clip3(-32768, 32767, (acc + (1 << (s2 - 1))) >> s2);
  }
}

/* ============================================================================================ */
/* TU descriptor: the subset of TComTU (TComTU.h/.cpp) that 4:2:0 intra coding needs             */
/* ============================================================================================ */
typedef struct {
  int cuZ, cuDepth, cuParts;
  int relZ, trDepth, log2, parts, section;
  int x, y;                 /* luma position inside the CTU */
  int cW;                   /* chroma block width handled by this TU (0: none here) */
  int cCodeAll, cTrDepth, cRelZ, cParts, cOff, cx, cy;
} TU;

static TU tu_root(int cuZ, int cuDepth)
{ /* TComTU::TComTU(pcCU, absPartIdxCU, cuDepth, 0), TComTU.cpp:48 */
  TU t; memset(&t, 0, sizeof(t));
  t.cuZ = cuZ; t.cuDepth = cuDepth; t.cuParts = 256 >> (2 * cuDepth);
  t.log2 = 6 - cuDepth; t.parts = t.cuParts;
  t.x = (Z2R[cuZ] & 15) * 4; t.y = (Z2R[cuZ] >> 4) * 4;
  t.cW = 1 << (t.log2 - 1); t.cCodeAll = 1; t.cParts = t.parts; t.cOff = cuZ * 4; t.cx = t.x >> 1; t.cy = t.y >> 1;
  return t;
}
static TU tu_child(const TU *p, int section, int processLast)
{ /* TComTU::TComTU(parent, bProcessLastOfLevel, QUAD_SPLIT) + nextSection, TComTU.cpp:88-185 */
  TU t = *p;
  t.log2 = p->log2 - 1; t.trDepth = p->trDepth + 1; t.parts = p->parts >> 2; if (t.parts < 1) t.parts = 1;
  t.relZ = p->relZ + section * t.parts; t.section = section;
  t.x = p->x + (section & 1) * (1 << t.log2); t.y = p->y + (section >> 1) * (1 << t.log2);
  if (t.log2 >= 3) {
    t.cW = 1 << (t.log2 - 1); t.cCodeAll = 1; t.cTrDepth = t.trDepth; t.cRelZ = t.relZ; t.cParts = t.parts;
    t.cOff = (t.cuZ + t.relZ) * 4; t.cx = t.x >> 1; t.cy = t.y >> 1;
  } else { /* 4x4 luma: the 4x4 chroma block of the parent is carried by one quadrant */
    t.cCodeAll = 0; t.cTrDepth = p->cTrDepth; t.cRelZ = t.relZ & ~3; t.cParts = t.parts * 4;
    t.cOff = p->cOff; t.cx = p->cx; t.cy = p->cy;
    t.cW = (section == (processLast ? 3 : 0)) ? 4 : 0;
  }
  return t;
}

/* ============================================================================================ */
/* neighbour helpers                                                                             */
/* ============================================================================================ */
static inline CtuMeta *meta_at(Enc *e, int x4, int y4, int *z)
{ *z = R2Z[((y4 & 15) << 4) | (x4 & 15)]; return e->meta + ((y4 >> 4) * e->wCtu + (x4 >> 4)); }

/* TComDataCU::getIntraDirPredictor, TComDataCU.cpp:1513-1586 (luma) */
static int intra_dir_predictor(Enc *e, int z, int preds[3], int *mode)
{
  int x4 = e->ctuX * 16 + (Z2R[z] & 15), y4 = e->ctuY * 16 + (Z2R[z] >> 4);
  int left = DC_IDX, above = DC_IDX, zz;
  if (x4 > 0) { CtuMeta *m = meta_at(e, x4 - 1, y4, &zz); left = (m->pred[zz] == MODE_INTRA) ? m->dirL[zz] : DC_IDX; }
  if ((y4 & 15) != 0) { CtuMeta *m = meta_at(e, x4, y4 - 1, &zz); above = (m->pred[zz] == MODE_INTRA) ? m->dirL[zz] : DC_IDX; }
  if (left == above) {
    if (mode) *mode = 1;
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = PLANAR_IDX; preds[1] = DC_IDX; preds[2] = VER_IDX; }
  } else {
    if (mode) *mode = 2;
    preds[0] = left; preds[1] = above;
    if (left && above) preds[2] = PLANAR_IDX; else preds[2] = (left + above) < 2 ? VER_IDX : DC_IDX;
  }
  return 3;
}

/* TComDataCU::getCtxSplitFlag, TComDataCU.cpp:1587-1601 */
static int ctx_split_flag(Enc *e, int z, int depth)
{
  int x4 = e->ctuX * 16 + (Z2R[z] & 15), y4 = e->ctuY * 16 + (Z2R[z] >> 4), ctx = 0, zz;
  if (x4 > 0) { CtuMeta *m = meta_at(e, x4 - 1, y4, &zz); ctx += m->depth[zz] > depth; }
  if (y4 > 0) { CtuMeta *m = meta_at(e, x4, y4 - 1, &zz); ctx += m->depth[zz] > depth; }
  return ctx;
}

/* ============================================================================================ */
/* intra reference samples (TComPattern.cpp:107-500) and prediction (TComPrediction.cpp:182-840) */
/* ============================================================================================ */
/* availability of one 4x4 unit seen from the TU whose RT / LB unit is given (TComDataCU.cpp:1244-1360) */
static int avail_above_right(Enc *e, int rtx4, int rty4, int k)
{ /* getPUAboveRightAdi: rt = absolute 4x4 position of the TU's top-right unit */
  if ((rtx4 + k) * 4 >= e->cfg.width) return 0;
  int cx = rtx4 & 15, cy = rty4 & 15;
  if (cx + k <= 15) {
    if (cy != 0) return R2Z[(cy << 4) | cx] > R2Z[((cy - 1) << 4) | (cx + k)];
    return rty4 > 0;                                   /* above CTU */
  }
  if (cy != 0) return 0;
  return rty4 > 0 && (rtx4 >> 4) < e->wCtu - 1;         /* above-right CTU */
}
static int avail_below_left(Enc *e, int lbx4, int lby4, int k)
{ /* getPUBelowLeftAdi */
  if ((lby4 + k) * 4 >= e->cfg.height) return 0;
  int cx = lbx4 & 15, cy = lby4 & 15;
  if (cy + k <= 15) {
    if (cx != 0) return R2Z[(cy << 4) | cx] > R2Z[((cy + k) << 4) | (cx - 1)];
    return lbx4 > 0;                                   /* left CTU */
  }
  return 0;
}

/* TComPrediction::initAdiPatternChType + fillReferenceSamples + smoothing.
 * comp: 0..2; (px,py): block position in the component plane (absolute); n: block size;
 * (x4,y4): absolute 4x4-unit position of the block's top-left luma unit; units: block size in units */
static void init_adi_pattern(Enc *e, int comp, int px, int py, int n, int x4, int y4, int units, int filter)
{
  const int uw = comp ? 2 : 4;                          /* samples per unit in this plane */
  const int total = 4 * units + 1;
  uint8_t flags[4 * 16 + 1];
  int num = 0;
  const int L = 2 * units;                              /* iLeftUnits */
  flags[L] = (x4 > 0 && y4 > 0); num += flags[L];
  for (int i = 0; i < units; i++) { flags[L + 1 + i] = (y4 > 0); num += flags[L + 1 + i]; }
  for (int k = 1; k <= units; k++) { int a = avail_above_right(e, x4 + units - 1, y4, k); flags[L + units + k] = (uint8_t)a; num += a; }
  for (int i = 0; i < units; i++) { flags[L - 1 - i] = (x4 > 0); num += flags[L - 1 - i]; }
  for (int k = 1; k <= units; k++) { int a = avail_below_left(e, x4, y4 + units - 1, k); flags[L - units - k] = (uint8_t)a; num += a; }

  const Pel *rec = e->rec[comp]; const int st = e->stride[comp];
  const int dc = 1 << (e->cfg.bit_depth - 1);
  Pel *top = e->refTop[0], *left = e->refLeft[0];
  const int n2 = 2 * n;
  if (num == 0) {
    for (int i = 0; i <= n2; i++) { top[i] = (Pel)dc; left[i] = (Pel)dc; }
  } else if (num == total) {
    for (int i = 0; i <= n2; i++) top[i] = rec[(py - 1) * st + px - 1 + i];
    left[0] = top[0];
    for (int i = 1; i <= n2; i++) left[i] = rec[(py - 1 + i) * st + px - 1];
  } else {
    /* line[]: 2n left samples bottom-to-top, uw copies of the top-left sample, 2n above samples */
    Pel line[5 * 64 + 8];
    const int nl = n2 + uw + n2;
    for (int i = 0; i < nl; i++) line[i] = (Pel)dc;
    if (flags[L]) for (int i = 0; i < uw; i++) line[n2 + i] = rec[(py - 1) * st + px - 1];
    for (int j = 0; j < L; j++)                       /* left & below-left, unit j counted downwards */
      if (flags[L - 1 - j]) for (int i = 0; i < uw; i++) line[n2 - 1 - (j * uw + i)] = rec[(py + j * uw + i) * st + px - 1];
    for (int j = 0; j < L; j++)                       /* above & above-right */
      if (flags[L + 1 + j]) for (int i = 0; i < uw; i++) line[n2 + uw + j * uw + i] = rec[(py - 1) * st + px + j * uw + i];
    /* padding (TComPattern.cpp:432-484); unit u occupies [u*uw, (u+1)*uw) in line[] */
    int cur = 0;
    if (!flags[0]) {
      int next = 1; while (next < total && !flags[next]) next++;
      Pel ref = line[next * uw];
      for (; cur < next; cur++) for (int i = 0; i < uw; i++) line[cur * uw + i] = ref;
    }
    for (; cur < total; cur++)
      if (!flags[cur]) { Pel ref = line[cur * uw - 1]; for (int i = 0; i < uw; i++) line[cur * uw + i] = ref; }
    for (int i = 0; i <= n2; i++) top[i] = line[n2 + uw - 1 + i];
    left[0] = top[0];
    for (int i = 1; i <= n2; i++) left[i] = line[n2 - i];
  }
  if (!filter) return;
  /* smoothing, TComPattern.cpp:180-283 */
  Pel *ft = e->refTop[1], *fl = e->refLeft[1];
  int strong = (comp == 0);                             /* SPS strong_intra_smoothing_enabled (cfg default 1) */
  const int bl = left[n2], tl = top[0], tr = top[n2];
  if (strong) {
    const int thr = 1 << (e->cfg.bit_depth - 5);
    int bilLeft = abs((bl + tl) - 2 * left[n]) < thr, bilAbove = abs((tl + tr) - 2 * top[n]) < thr;
    if (n < 32 || !bilLeft || !bilAbove) strong = 0;
  }
  fl[n2] = left[n2]; ft[n2] = top[n2];
  if (strong) {
    const int shift = ilog2(n) + 1;
    for (int i = 1; i < n2; i++) fl[n2 - i] = (Pel)(((n2 - i) * bl + i * tl + n) >> shift);
    ft[0] = fl[0] = top[0];
    for (int i = 1; i < n2; i++) ft[i] = (Pel)(((n2 - i) * tl + i * tr + n) >> shift);
  } else {
    for (int i = n2 - 1; i >= 1; i--) fl[i] = (Pel)((left[i + 1] + 2 * left[i] + left[i - 1] + 2) >> 2);
    ft[0] = fl[0] = (Pel)((left[1] + 2 * top[0] + top[1] + 2) >> 2);
    for (int i = 1; i < n2; i++) ft[i] = (Pel)((top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2);
  }
}

/* TComPrediction::filteringIntraReferenceSamples, TComPattern.cpp:514-540 */
static int use_filtered_refs(int comp, int mode, int n)
{
  if (comp != 0) return 0;                               /* 4:2:0 chroma is never smoothed */
  if (mode == DC_IDX) return 0;
  int d1 = abs(mode - HOR_IDX), d2 = abs(mode - VER_IDX);
  return (d1 < d2 ? d1 : d2) > INTRA_FILTER[ilog2(n) - 2];
}

/* TComPrediction::predIntraAng (+xPredIntraPlanar/xPredIntraAng/xDCPredFiltering), TComPrediction.cpp:182-840 */
static void pred_intra(Enc *e, int comp, int mode, int n, int filtered, Pel *dst, int ds)
{
  const Pel *top = e->refTop[filtered], *left = e->refLeft[filtered];   /* [0] is the corner */
  const int bd = e->cfg.bit_depth;
  if (mode == PLANAR_IDX) {
    int l2 = ilog2(n), topRow[64], bottomRow[64], leftCol[64], rightCol[64];
    int bottomLeft = left[n + 1], topRight = top[n + 1];
    for (int k = 0; k < n; k++) { bottomRow[k] = bottomLeft - top[k + 1]; topRow[k] = top[k + 1] << l2; rightCol[k] = topRight - left[k + 1]; leftCol[k] = left[k + 1] << l2; }
    for (int y = 0; y < n; y++) {
      int hor = leftCol[y] + n;
      for (int x = 0; x < n; x++) { hor += rightCol[y]; topRow[x] += bottomRow[x]; dst[y * ds + x] = (Pel)((hor + topRow[x]) >> (l2 + 1)); }
    }
    return;
  }
  if (mode == DC_IDX) {
    int sum = 0; for (int i = 0; i < n; i++) sum += top[i + 1] + left[i + 1];
    Pel dc = (Pel)((sum + n) / (n + n));
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[y * ds + x] = dc;
    if (comp == 0 && n <= 16) { /* xDCPredFiltering */
      dst[0] = (Pel)((top[1] + left[1] + 2 * dst[0] + 2) >> 2);
      for (int x = 1; x < n; x++) dst[x] = (Pel)((top[x + 1] + 3 * dst[x] + 2) >> 2);
      for (int y = 1; y < n; y++) dst[y * ds] = (Pel)((left[y + 1] + 3 * dst[y * ds] + 2) >> 2);
    }
    return;
  }
  static const int angTable[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
  static const int invAngTable[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER_IDX : -(mode - HOR_IDX);
  const int absAng = angTable[abs(angMode)], invAngle = invAngTable[abs(angMode)];
  const int angle = angMode < 0 ? -absAng : absAng;
  Pel refBuf[2][3 * 64 + 4]; Pel *refMain, *refSide;
  if (angle < 0) {
    refMain = refBuf[0] + 64; refSide = refBuf[1] + 64;
    for (int i = 0; i <= n; i++) { refMain[i] = isVer ? top[i] : left[i]; refSide[i] = isVer ? left[i] : top[i]; }
    int invAngleSum = 128;                                /* extend the main reference to the left */
    for (int k = -1; k > (n * angle) >> 5; k--) { invAngleSum += invAngle; refMain[k] = refSide[invAngleSum >> 8]; }
  } else {
    for (int i = 0; i <= 2 * n; i++) { refBuf[0][i] = isVer ? top[i] : left[i]; refBuf[1][i] = isVer ? left[i] : top[i]; }
    refMain = refBuf[0]; refSide = refBuf[1];
  }
  Pel tmp[64 * 64]; Pel *pd = isVer ? dst : tmp; const int ps = isVer ? ds : 64;
  if (angle == 0) {
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) pd[y * ps + x] = refMain[x + 1];
    if (comp == 0 && n <= 16)
      for (int y = 0; y < n; y++) pd[y * ps] = (Pel)clip3(0, (1 << bd) - 1, pd[y * ps] + ((refSide[y + 1] - refSide[0]) >> 1));
  } else {
    for (int y = 0, deltaPos = angle; y < n; y++, deltaPos += angle) {
      const int di = deltaPos >> 5, df = deltaPos & 31;
      if (df) for (int x = 0; x < n; x++) pd[y * ps + x] = (Pel)(((32 - df) * refMain[x + di + 1] + df * refMain[x + di + 2] + 16) >> 5);
      else for (int x = 0; x < n; x++) pd[y * ps + x] = refMain[x + di + 1];
    }
  }
  if (!isVer) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) dst[x * ds + y] = pd[y * ps + x];
}

/* ============================================================================================ */
/* coefficient coding parameters                                                                 */
/* ============================================================================================ */
/* TComDataCU::getCoefScanIdx, TComDataCU.cpp:3340-3380 */
static int coef_scan_idx(const CtuMeta *m, int z, int n, int comp)
{
  if (m->pred[z] != MODE_INTRA) return SCAN_DIAG;     /* getMDCSScanOrder... only intra CUs scan mode-dependently, TComTU.cpp / TComCodingStatistics */
  if (n > (comp ? 4 : 8)) return SCAN_DIAG;
  int dir = comp ? m->dirC[z] : m->dirL[z];
  if (dir == DM_CHROMA_IDX) dir = m->dirL[z & ~3];
  if (abs(dir - VER_IDX) <= 4) return SCAN_HOR;
  if (abs(dir - HOR_IDX) <= 4) return SCAN_VER;
  return SCAN_DIAG;
}
static inline int first_sig_ctx(int n, int scanType, int chroma)
{ /* getTUEntropyCodingParameters, TComChromaFormat.cpp:75-130 */
  if (n == 4) return 0;
  if (n == 8) return 9 + ((scanType != SCAN_DIAG && !chroma) ? 6 : 0);
  return chroma ? 12 : 21;
}
/* TComTrQuant::getSigCtxInc, TComTrQuant.cpp:2548-2640 */
static int sig_ctx_inc(int pattern, int firstCtx, int blkPos, int log2n, int chroma)
{
  const int posY = blkPos >> log2n, posX = blkPos - (posY << log2n);
  if (posX + posY == 0) return 0;
  int offset;
  if (log2n == 2) offset = CTX_IND_MAP_4x4[4 * posY + posX];
  else {
    int cnt, xs = posX & 3, ys = posY & 3;
    switch (pattern) {
      case 0: cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2); break;
      case 1: cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2); break;
      case 2: cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2); break;
      default: cnt = 2; break;
    }
    const int notFirst = ((posX >> 2) + (posY >> 2)) > 0;
    offset = ((notFirst && !chroma) ? 3 : 0) + cnt;
  }
  return firstCtx + offset;
}
static inline int pattern_sig_ctx(const uint8_t *cgFlag, int cgx, int cgy, int wg)
{ /* TComTrQuant::calcPatternSigCtx, TComTrQuant.cpp:2522-2535 */
  if (wg <= 1) return 0;
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgFlag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgFlag[(cgy + 1) * wg + cgx] != 0;
  return r + (l << 1);
}
static inline int sig_cg_ctx(const uint8_t *cgFlag, int cgx, int cgy, int wg)
{ /* TComTrQuant::getSigCoeffGroupCtxInc, TComTrQuant.cpp:2872-2886 */
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgFlag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgFlag[(cgy + 1) * wg + cgx] != 0;
  return (r + l) != 0;
}
static inline int ctx_set_index(int chroma, int subset, int gt1)
{ /* getContextSetIndex, TComChromaFormat.h:243 */
  return (chroma ? 4 : 0) + ((!chroma && subset > 0) ? 2 : 0) + (gt1 ? 1 : 0);
}
static inline void last_ctx_params(int chroma, int n, int *off, int *shift)
{ /* getLastSignificantContextParameters, TComChromaFormat.h:211 */
  int cv = ilog2(n) - 2;
  *off = chroma ? 0 : (cv * 3 + ((cv + 1) >> 2));
  *shift = chroma ? cv : ((cv + 3) >> 2);
}

/* ============================================================================================ */
/* RDOQ (TComTrQuant::xRateDistOptQuant, TComTrQuant.cpp:1974-2511)                              */
/* ============================================================================================ */
static inline int ic_rate(const Cabac *c, uint32_t absLevel, int ctxOne, int ctxAbs, int goRice, int c1Idx, int c2Idx)
{ /* xGetICRate, TComTrQuant.cpp:2725-2800 (useLimitedPrefixLength = 0) */
  int rate = 32768;
  uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < (3u << goRice)) { length = symbol >> goRice; rate += (length + 1 + goRice) << 15; }
    else {
      length = goRice; symbol -= (3u << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (3 + length + 1 - goRice + length) << 15;
    }
    if (c1Idx < 8) { rate += ebits(c, C_ONE + ctxOne, 1); if (c2Idx < 1) rate += ebits(c, C_ABS + ctxAbs, 1); }
  } else if (absLevel == 1) rate += ebits(c, C_ONE + ctxOne, 0);
  else if (absLevel == 2) { rate += ebits(c, C_ONE + ctxOne, 1); rate += ebits(c, C_ABS + ctxAbs, 0); }
  else rate = 0;
  return rate;
}

/* returns absSum; dst receives signed levels.  cbfCtx: context index inside C_QT_CBF */
static int rdoq(Enc *e, const TCoeff *src, TCoeff *dst, int n, int comp, int scanType, int tskip, int cbfCtx)
{
  const Cabac *cb = &e->cur;
  const int chroma = comp != 0, log2n = ilog2(n), bd = e->cfg.bit_depth;
  const double lambda = chroma ? e->lambdaC : e->lambda;
  const int transformShift = 15 - bd - log2n;
  const int qBits = 14 + e->qpPer[comp] + transformShift;
  const int quantCoef = QUANT_SCALES[e->qpRem[comp]];
  /* setErrScaleCoeff, TComTrQuant.cpp:2933-2956 */
  double errScale = (double)(1 << 15);
  errScale = errScale * pow(2.0, -2.0 * transformShift);
  errScale = errScale / quantCoef / quantCoef / (double)(1 << (2 * (bd - 8)));
  const int numCoef = n * n, wg = n >> 2, cgNum = numCoef >> 4;
  const uint16_t *scan = SCAN[scanType][log2n - 2], *scanCG = SCANCG[scanType][log2n - 2];
  const int firstCtx = first_sig_ctx(n, scanType, chroma);
  const int sigOff = C_SIG + (chroma ? 28 : 0);
  double costCoeff[1024], costSig[1024], costCoeff0[1024];
  int rateIncUp[1024], rateIncDown[1024], sigRateDelta[1024]; TCoeff deltaU[1024];
  double costCGSig[64]; uint8_t cgFlag[64];
  memset(costCoeff, 0, sizeof(double) * numCoef); memset(costSig, 0, sizeof(double) * numCoef);
  memset(rateIncUp, 0, sizeof(int) * numCoef); memset(rateIncDown, 0, sizeof(int) * numCoef);
  memset(sigRateDelta, 0, sizeof(int) * numCoef); memset(deltaU, 0, sizeof(TCoeff) * numCoef);
  memset(costCGSig, 0, sizeof(costCGSig)); memset(cgFlag, 0, sizeof(cgFlag));
  (void)tskip;
  double blockUncodedCost = 0, baseCost = 0;
  int cgLastScanPos = -1, lastScanPos = -1, ctxSet = 0, c1 = 1, c2 = 0, c1Idx = 0, c2Idx = 0, goRice = 0;
  for (int cgScanPos = cgNum - 1; cgScanPos >= 0; cgScanPos--) {
    const int cgBlkPos = scanCG[cgScanPos], cgy = cgBlkPos / wg, cgx = cgBlkPos - cgy * wg;
    double sigCost = 0, sigCost0 = 0, codedLevelAndDist = 0, uncodedDist = 0; int nnzBeforePos0 = 0;
    const int pattern = pattern_sig_ctx(cgFlag, cgx, cgy, wg);
    for (int posInCG = 15; posInCG >= 0; posInCG--) {
      const int scanPos = cgScanPos * 16 + posInCG, blkPos = scan[scanPos];
      const int64_t tmpLevel = (int64_t)abs(src[blkPos]) * quantCoef;
      const int64_t cap = 2147483647LL - (1LL << (qBits - 1));
      const int32_t levelDouble = (int32_t)(tmpLevel < cap ? tmpLevel : cap);
      uint32_t maxAbsLevel = (uint32_t)((levelDouble + (1 << (qBits - 1))) >> qBits);
      if (maxAbsLevel > 32767u) maxAbsLevel = 32767u;
      const double err = (double)levelDouble;
      costCoeff0[scanPos] = err * err * errScale;
      blockUncodedCost += costCoeff0[scanPos];
      dst[blkPos] = (TCoeff)maxAbsLevel;
      if (maxAbsLevel > 0 && lastScanPos < 0) { lastScanPos = scanPos; ctxSet = ctx_set_index(chroma, scanPos >> 4, 0); cgLastScanPos = cgScanPos; }
      if (lastScanPos >= 0) {
        const int ctxOne = 4 * ctxSet + c1, ctxAbs = ctxSet;
        uint32_t level = 0; int ctxSig = 0;
        const int isLast = (scanPos == lastScanPos);
        if (!isLast) ctxSig = sigOff + sig_ctx_inc(pattern, firstCtx, blkPos, log2n, chroma);
        { /* xGetCodedLevel, TComTrQuant.cpp:2660-2715 */
          double currCostSig = 0; int done = 0;
          if (!isLast && maxAbsLevel < 3) {
            costSig[scanPos] = lambda * (double)ebits(cb, ctxSig, 0);
            costCoeff[scanPos] = costCoeff0[scanPos] + costSig[scanPos];
            if (maxAbsLevel == 0) done = 1;
          } else costCoeff[scanPos] = MAX_DOUBLE;
          if (!done) {
            if (!isLast) currCostSig = lambda * (double)ebits(cb, ctxSig, 1);
            const uint32_t minAbs = maxAbsLevel > 1 ? maxAbsLevel - 1 : 1;
            for (int al = (int)maxAbsLevel; al >= (int)minAbs; al--) {
              const double de = (double)(levelDouble - (int32_t)((uint32_t)al << qBits));
              double cc = de * de * errScale + lambda * (double)ic_rate(cb, (uint32_t)al, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
              cc += currCostSig;
              if (cc < costCoeff[scanPos]) { level = (uint32_t)al; costCoeff[scanPos] = cc; costSig[scanPos] = currCostSig; }
            }
          }
        }
        if (!isLast) sigRateDelta[blkPos] = ebits(cb, ctxSig, 1) - ebits(cb, ctxSig, 0);
        deltaU[blkPos] = (TCoeff)((levelDouble - (int32_t)(level << qBits)) >> (qBits - 8));
        if (level > 0) {
          const int rateNow = ic_rate(cb, level, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
          rateIncUp[blkPos] = ic_rate(cb, level + 1, ctxOne, ctxAbs, goRice, c1Idx, c2Idx) - rateNow;
          rateIncDown[blkPos] = ic_rate(cb, level - 1, ctxOne, ctxAbs, goRice, c1Idx, c2Idx) - rateNow;
        } else rateIncUp[blkPos] = ebits(cb, C_ONE + ctxOne, 0);
        dst[blkPos] = (TCoeff)level;
        baseCost += costCoeff[scanPos];
        const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel && level > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
        if ((scanPos % 16 == 0) && scanPos > 0) {
          ctxSet = ctx_set_index(chroma, (scanPos - 1) >> 4, c1 == 0);
          c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0;
        }
      } else baseCost += costCoeff0[scanPos];
      sigCost += costSig[scanPos];
      if (posInCG == 0) sigCost0 = costSig[scanPos];
      if (dst[blkPos]) {
        cgFlag[cgBlkPos] = 1;
        codedLevelAndDist += costCoeff[scanPos] - costSig[scanPos];
        uncodedDist += costCoeff0[scanPos];
        if (posInCG != 0) nnzBeforePos0++;
      }
    }
    if (cgLastScanPos >= 0) {
      if (cgScanPos) {
        if (cgFlag[cgBlkPos] == 0) {
          const int ctx = C_SIG_CG + (chroma ? 2 : 0) + sig_cg_ctx(cgFlag, cgx, cgy, wg);
          baseCost += lambda * (double)ebits(cb, ctx, 0) - sigCost;
          costCGSig[cgScanPos] = lambda * (double)ebits(cb, ctx, 0);
        } else if (cgScanPos < cgLastScanPos) {
          if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
          double costZeroCG = baseCost;
          const int ctx = C_SIG_CG + (chroma ? 2 : 0) + sig_cg_ctx(cgFlag, cgx, cgy, wg);
          baseCost += lambda * (double)ebits(cb, ctx, 1);
          costZeroCG += lambda * (double)ebits(cb, ctx, 0);
          costCGSig[cgScanPos] = lambda * (double)ebits(cb, ctx, 1);
          costZeroCG += uncodedDist; costZeroCG -= codedLevelAndDist; costZeroCG -= sigCost;
          if (costZeroCG < baseCost) {
            cgFlag[cgBlkPos] = 0; baseCost = costZeroCG;
            costCGSig[cgScanPos] = lambda * (double)ebits(cb, ctx, 0);
            for (int posInCG = 15; posInCG >= 0; posInCG--) {
              const int scanPos = cgScanPos * 16 + posInCG, blkPos = scan[scanPos];
              if (dst[blkPos]) { dst[blkPos] = 0; costCoeff[scanPos] = costCoeff0[scanPos]; costSig[scanPos] = 0; }
            }
          }
        }
      } else cgFlag[cgBlkPos] = 1;
    }
  }
  if (lastScanPos < 0) return 0;
  double bestCost;
  { /* intra: per-TU cbf context (TComTrQuant.cpp:2310-2316) */
    bestCost = blockUncodedCost + lambda * (double)ebits(cb, C_QT_CBF + cbfCtx, 0);
    baseCost += lambda * (double)ebits(cb, C_QT_CBF + cbfCtx, 1);
  }
  int bestLastIdxP1 = 0, foundLast = 0;
  int lastOff, lastShift; last_ctx_params(chroma, n, &lastOff, &lastShift);
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0 && !foundLast; cgScanPos--) {
    const int cgBlkPos = scanCG[cgScanPos];
    baseCost -= costCGSig[cgScanPos];
    if (!cgFlag[cgBlkPos]) continue;
    for (int posInCG = 15; posInCG >= 0; posInCG--) {
      const int scanPos = cgScanPos * 16 + posInCG;
      if (scanPos > lastScanPos) continue;
      const int blkPos = scan[scanPos];
      if (dst[blkPos]) {
        int posY = blkPos >> log2n, posX = blkPos - (posY << log2n);
        if (scanType == SCAN_VER) { int t = posX; posX = posY; posY = t; }
        /* xGetRateLast, TComTrQuant.cpp:2815-2832 with estLastSignificantPositionBit, TEncSbac.cpp:1846-1892 */
        const int gx = GROUP_IDX[posX], gy = GROUP_IDX[posY];
        int bx = 0, by = 0;
        for (int c = 0; c < gx; c++) bx += ebits(cb, C_LASTX + (chroma ? 15 : 0) + lastOff + (c >> lastShift), 1);
        if (gx < GROUP_IDX[n - 1]) bx += ebits(cb, C_LASTX + (chroma ? 15 : 0) + lastOff + (gx >> lastShift), 0);
        for (int c = 0; c < gy; c++) by += ebits(cb, C_LASTY + (chroma ? 15 : 0) + lastOff + (c >> lastShift), 1);
        if (gy < GROUP_IDX[n - 1]) by += ebits(cb, C_LASTY + (chroma ? 15 : 0) + lastOff + (gy >> lastShift), 0);
        double cst = (double)(bx + by);
        if (gx > 3) cst += 32768.0 * (double)((gx - 2) >> 1);
        if (gy > 3) cst += 32768.0 * (double)((gy - 2) >> 1);
        const double costLast = lambda * cst;
        const double totalCost = baseCost + costLast - costSig[scanPos];
        if (totalCost < bestCost) { bestLastIdxP1 = scanPos + 1; bestCost = totalCost; }
        if (dst[blkPos] > 1) { foundLast = 1; break; }
        baseCost -= costCoeff[scanPos]; baseCost += costCoeff0[scanPos];
      } else baseCost -= costSig[scanPos];
    }
  }
  int absSum = 0;
  for (int sp = 0; sp < bestLastIdxP1; sp++) { const int bp = scan[sp]; const TCoeff lv = dst[bp]; absSum += lv; dst[bp] = src[bp] < 0 ? -lv : lv; }
  for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) dst[scan[sp]] = 0;
  /* sign bit hiding, TComTrQuant.cpp:2380-2510 */
  if (absSum >= 2) {
    const double invQ = (double)INV_QUANT_SCALES[e->qpRem[comp]];
    const int64_t rdFactor = (int64_t)(invQ * invQ * (double)(1 << (2 * e->qpPer[comp])) / lambda / 16 / (double)(1 << (2 * (bd - 8))) + 0.5);
    int lastCG = -1;
    for (int subSet = (numCoef - 1) >> 4; subSet >= 0; subSet--) {
      const int subPos = subSet << 4; int firstNZ = 16, lastNZ = -1, sum = 0, k;
      for (k = 15; k >= 0; --k) if (dst[scan[k + subPos]]) { lastNZ = k; break; }
      for (k = 0; k < 16; k++) if (dst[scan[k + subPos]]) { firstNZ = k; break; }
      for (k = firstNZ; k <= lastNZ; k++) sum += dst[scan[k + subPos]];
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4) {
        const uint32_t signbit = dst[scan[subPos + firstNZ]] > 0 ? 0 : 1;
        if (signbit != (uint32_t)(sum & 1)) {
          int64_t minCostInc = INT64_MAX, curCost = INT64_MAX; int minPos = -1, finalChange = 0, curChange = 0;
          for (k = (lastCG == 1 ? lastNZ : 15); k >= 0; --k) {
            const int bp = scan[k + subPos];
            if (dst[bp] != 0) {
              int64_t costUp = rdFactor * (-deltaU[bp]) + rateIncUp[bp];
              int64_t costDown = rdFactor * (deltaU[bp]) + rateIncDown[bp] - ((abs(dst[bp]) == 1) ? sigRateDelta[bp] : 0);
              if (lastCG == 1 && lastNZ == k && abs(dst[bp]) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else { curChange = -1; if (k == firstNZ && abs(dst[bp]) == 1) curCost = INT64_MAX; else curCost = costDown; }
            } else {
              curCost = rdFactor * (-(abs(deltaU[bp]))) + (1 << 15) + rateIncUp[bp] + sigRateDelta[bp];
              curChange = 1;
              if (k < firstNZ) { const uint32_t thissign = src[bp] >= 0 ? 0 : 1; if (thissign != signbit) curCost = INT64_MAX; }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = bp; }
          }
          if (dst[minPos] == 32767 || dst[minPos] == -32768) finalChange = -1;
          if (src[minPos] >= 0) dst[minPos] += finalChange; else dst[minPos] -= finalChange;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return absSum;
}

/* TComTrQuant::xDeQuant (flat scaling), TComTrQuant.cpp:1276-1312 */
static void dequant(const Enc *e, const TCoeff *q, TCoeff *out, int n, int comp)
{
  const int transformShift = 15 - e->cfg.bit_depth - ilog2(n);
  const int rightShift = 6 - (transformShift + e->qpPer[comp]);
  const int scale = INV_QUANT_SCALES[e->qpRem[comp]];
  const int num = n * n;
  int tgt = 25 + rightShift; if (tgt > 16) tgt = 16;
  const int imin = -(1 << (tgt - 1)), imax = (1 << (tgt - 1)) - 1;
  if (rightShift > 0) {
    const int add = 1 << (rightShift - 1);
    for (int i = 0; i < num; i++) { const int c = clip3(imin, imax, q[i]); out[i] = clip3(-32768, 32767, (c * scale + add) >> rightShift); }
  } else {
    const int ls = -rightShift;
    for (int i = 0; i < num; i++) { const int c = clip3(imin, imax, q[i]); out[i] = clip3(-32768, 32767, (int)((unsigned)(c * scale) << ls)); }
  }
}

/* ============================================================================================ */
/* syntax element coding on the estimator (TEncSbac.cpp)                                         */
/* ============================================================================================ */
/* TEncSbac::codeCoeffNxN, TEncSbac.cpp:1172-1525 (+codeLastSignificantXY :1106, xWriteCoefRemainExGolomb :337) */
static void code_coeff_nxn(Enc *e, Cabac *c, const TCoeff *coef, int n, int comp, int scanType, int tskipFlag)
{
  const int chroma = comp != 0, log2n = ilog2(n), wg = n >> 2;
  if (n == 4) enc_bin(c, C_TSKIP + chroma, tskipFlag);          /* codeTransformSkipFlags, TEncSbac.cpp:988 */
  const uint16_t *scan = SCAN[scanType][log2n - 2], *scanCG = SCANCG[scanType][log2n - 2];
  int numSig = 0; for (int i = 0; i < n * n; i++) numSig += coef[i] != 0;
  uint8_t cgFlag[64]; memset(cgFlag, 0, sizeof(cgFlag));
  int scanPosLast = -1, posLast;
  do {
    posLast = scan[++scanPosLast];
    if (coef[posLast] != 0) { const int py = posLast >> log2n, px = posLast - (py << log2n); cgFlag[wg * (py >> 2) + (px >> 2)] = 1; numSig--; }
  } while (numSig > 0);
  { /* last position */
    int py = posLast >> log2n, px = posLast - (py << log2n);
    if (scanType == SCAN_VER) { int t = px; px = py; py = t; }
    const int gx = GROUP_IDX[px], gy = GROUP_IDX[py];
    int off, shift; last_ctx_params(chroma, n, &off, &shift);
    const int bxc = C_LASTX + (chroma ? 15 : 0) + off, byc = C_LASTY + (chroma ? 15 : 0) + off;
    int k;
    for (k = 0; k < gx; k++) enc_bin(c, bxc + (k >> shift), 1);
    if (gx < GROUP_IDX[n - 1]) enc_bin(c, bxc + (k >> shift), 0);
    for (k = 0; k < gy; k++) enc_bin(c, byc + (k >> shift), 1);
    if (gy < GROUP_IDX[n - 1]) enc_bin(c, byc + (k >> shift), 0);
    if (gx > 3) enc_epv(c, (uint32_t)(px - MIN_IN_GROUP[gx]), (gx - 2) >> 1);
    if (gy > 3) enc_epv(c, (uint32_t)(py - MIN_IN_GROUP[gy]), (gy - 2) >> 1);
  }
  const int firstCtx = first_sig_ctx(n, scanType, chroma), sigOff = C_SIG + (chroma ? 28 : 0);
  const int lastScanSet = scanPosLast >> 4;
  uint32_t c1 = 1; int scanPosSig = scanPosLast;
  for (int subSet = lastScanSet; subSet >= 0; subSet--) {
    int numNonZero = 0; const int subPos = subSet << 4; uint32_t goRice = 0;
    int absCoeff[16]; int lastNZ = -1, firstNZ = 16; uint32_t signs = 0; int escape = 0;
    if (scanPosSig == scanPosLast) {
      absCoeff[0] = abs(coef[posLast]); signs = coef[posLast] < 0; numNonZero = 1; lastNZ = scanPosSig; firstNZ = scanPosSig; scanPosSig--;
    }
    const int cgBlkPos = scanCG[subSet], cgy = cgBlkPos / wg, cgx = cgBlkPos - cgy * wg;
    if (subSet == lastScanSet || subSet == 0) cgFlag[cgBlkPos] = 1;
    else enc_bin(c, C_SIG_CG + (chroma ? 2 : 0) + sig_cg_ctx(cgFlag, cgx, cgy, wg), cgFlag[cgBlkPos] != 0);
    if (cgFlag[cgBlkPos]) {
      const int pattern = pattern_sig_ctx(cgFlag, cgx, cgy, wg);
      for (; scanPosSig >= subPos; scanPosSig--) {
        const int blkPos = scan[scanPosSig]; const int sig = coef[blkPos] != 0;
        if (scanPosSig > subPos || subSet == 0 || numNonZero)
          enc_bin(c, sigOff + sig_ctx_inc(pattern, firstCtx, blkPos, log2n, chroma), sig);
        if (sig) {
          absCoeff[numNonZero] = abs(coef[blkPos]); signs = 2 * signs + (coef[blkPos] < 0); numNonZero++;
          if (lastNZ == -1) lastNZ = scanPosSig;
          firstNZ = scanPosSig;
        }
      }
    } else scanPosSig = subPos - 1;
    if (numNonZero > 0) {
      const int signHidden = (lastNZ - firstNZ >= 4);
      const int ctxSet = ctx_set_index(chroma, subSet, c1 == 0);
      c1 = 1;
      const int numC1 = numNonZero < 8 ? numNonZero : 8; int firstC2 = -1;
      for (int idx = 0; idx < numC1; idx++) {
        const int sym = absCoeff[idx] > 1;
        enc_bin(c, C_ONE + 4 * ctxSet + (int)c1, sym);
        if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = idx; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && firstC2 != -1) { const int sym = absCoeff[firstC2] > 2; enc_bin(c, C_ABS + ctxSet, sym); if (sym) escape = 1; }
      escape = escape || (numNonZero > 8);
      if (signHidden) enc_epv(c, signs >> 1, numNonZero - 1); else enc_epv(c, signs, numNonZero);
      int firstCoeff2 = 1;
      if (escape)
        for (int idx = 0; idx < numNonZero; idx++) {
          const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (absCoeff[idx] >= baseLevel) {
            uint32_t sym = (uint32_t)(absCoeff[idx] - baseLevel);
            if (sym < (3u << goRice)) { const uint32_t len = sym >> goRice; enc_epv(c, (1u << (len + 1)) - 2, (int)len + 1); enc_epv(c, sym & ((1u << goRice) - 1), (int)goRice); }
            else { uint32_t len = goRice; sym -= (3u << goRice); while (sym >= (1u << len)) sym -= (1u << (len++));
                   enc_epv(c, (1u << (3 + len + 1 - goRice)) - 2, (int)(3 + len + 1 - goRice)); enc_epv(c, sym, (int)len); }
            if ((uint32_t)absCoeff[idx] > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
          }
          if (absCoeff[idx] >= 2) firstCoeff2 = 0;
        }
    }
  }
}

/* TEncSbac::codeIntraDirLumaAng, TEncSbac.cpp:636-690: parts = 1, or 4 for the final NxN CU header */
static void code_intra_dir_luma(Enc *e, Cabac *c, int z, int multiple)
{
  const CtuMeta *m = e->cm;
  const int partNum = multiple ? (m->part[z] == SIZE_NxN ? 4 : 1) : 1;
  const int partOffset = (256 >> (m->depth[z] << 1)) >> 2;
  int dir[4], preds[4][3], predIdx[4] = {-1, -1, -1, -1};
  for (int j = 0; j < partNum; j++) {
    dir[j] = m->dirL[z + partOffset * j];
    intra_dir_predictor(e, z + partOffset * j, preds[j], NULL);
    for (int i = 0; i < 3; i++) if (dir[j] == preds[j][i]) predIdx[j] = i;
    enc_bin(c, C_INTRA_LUMA, predIdx[j] != -1);
  }
  for (int j = 0; j < partNum; j++) {
    if (predIdx[j] != -1) { if (predIdx[j]) enc_epv(c, 2u | (uint32_t)(predIdx[j] - 1), 2); else enc_epv(c, 0, 1); }
    else {
      int *p = preds[j], d = dir[j], t;
      if (p[0] > p[1]) { t = p[0]; p[0] = p[1]; p[1] = t; }
      if (p[0] > p[2]) { t = p[0]; p[0] = p[2]; p[2] = t; }
      if (p[1] > p[2]) { t = p[1]; p[1] = p[2]; p[2] = t; }
      for (int i = 2; i >= 0; i--) d = d > p[i] ? d - 1 : d;
      enc_epv(c, (uint32_t)d, 5);
    }
  }
}
/* TEncSbac::codeIntraDirChroma, TEncSbac.cpp:692-718 */
static void code_intra_dir_chroma(Enc *e, Cabac *c, int z)
{
  if (e->cm->dirC[z] == DM_CHROMA_IDX) enc_bin(c, C_CHROMA_PRED, 0);
  else {
    enc_bin(c, C_CHROMA_PRED, 1);
    int list[4] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX}, idx = 0;       /* getAllowedChromaDir, TComDataCU.cpp:1486: the entry equal to the luma mode becomes 34 */
    const int luma = e->cm->dirL[z];
    for (int i = 0; i < 4; i++) if (list[i] == luma) list[i] = 34;
    for (int i = 0; i < 4; i++) if (list[i] == e->cm->dirC[z]) { idx = i; break; }
    enc_epv(c, (uint32_t)idx, 2);
  }
}
/* TEncSbac::codeQtCbf, TEncSbac.cpp:911-960 (square TUs only) */
static void code_qt_cbf(Enc *e, Cabac *c, const TU *t, int comp, int lowestLevel)
{
  const int z = t->cuZ + (comp ? t->cRelZ : t->relZ);
  const int ctx = comp ? t->trDepth : (t->trDepth == 0 ? 1 : 0);
  const int width = comp ? (1 << (t->log2 - 1)) : (1 << t->log2);
  const int canQuadSplit = width >= 8;
  const int lowestTUDepth = t->trDepth + ((!lowestLevel && !canQuadSplit) ? 1 : 0);
  enc_bin(c, C_QT_CBF + (comp ? 5 : 0) + ctx, (e->cm->cbf[comp][z] >> lowestTUDepth) & 1);
}

/* ============================================================================================ */
/* search-time bit counting (TEncSearch.cpp:856-1070)                                            */
/* ============================================================================================ */
static inline int tr_min_size_in_cu(int cuLog2, int nxn)
{ /* TComDataCU::getQuadtreeTULog2MinSizeInCU, TComDataCU.cpp:1618-1643 (max depth intra 3, TU log2 2..5) */
  const int maxDepth = 3;
  if (cuLog2 < 2 + maxDepth - 1 + nxn) return 2;
  int v = cuLog2 - (maxDepth - 1 + nxn);
  return v > 5 ? 5 : v;
}

/* xEncSubdivCbfQT, TEncSearch.cpp:856-921 */
static void enc_subdiv_cbf_qt(Enc *e, const TU *t, int bLuma, int bChroma)
{
  const CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  const int subdiv = m->tr[z] > t->trDepth;
  const int nxn = m->part[t->cuZ] == SIZE_NxN;
  if (nxn && t->trDepth == 0) { }
  else if (t->log2 > 5) { }
  else if (t->log2 == 2) { }
  else if (t->log2 == tr_min_size_in_cu(6 - t->cuDepth, nxn)) { }
  else if (bLuma) enc_bin(&e->cur, C_SUBDIV + (5 - t->log2), subdiv);
  if (bChroma)
    for (int comp = 1; comp < 3; comp++)
      if (t->cCodeAll && (t->trDepth == 0 || ((m->cbf[comp][z] >> (t->trDepth - 1)) & 1)))
        code_qt_cbf(e, &e->cur, t, comp, subdiv == 0);
  if (subdiv) { for (int s = 0; s < 4; s++) { TU ch = tu_child(t, s, 0); enc_subdiv_cbf_qt(e, &ch, bLuma, bChroma); } }
  else if (bLuma) code_qt_cbf(e, &e->cur, t, 0, 1);
}
/* xEncCoeffQT, TEncSearch.cpp:926-960 (bRealCoeff = false: coefficients come from the QT layer buffers) */
static void enc_coeff_qt(Enc *e, const TU *t, int comp)
{
  const CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  if (m->tr[z] > t->trDepth) { for (int s = 0; s < 4; s++) { TU ch = tu_child(t, s, 0); enc_coeff_qt(e, &ch, comp); } return; }
  if (comp && !t->cW) return;
  const int layer = 5 - t->log2;
  /* TEncEntropy::encodeCoeffNxN, TEncEntropy.cpp:683: cbf of the luma-style partition at the luma depth */
  if (!((m->cbf[comp][z] >> t->trDepth) & 1)) return;
  const int n = comp ? t->cW : (1 << t->log2);
  const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
  const TCoeff *coef = e->qtCoef[comp][layer] + (comp ? t->cOff : (t->cuZ + t->relZ) * 16);
  code_coeff_nxn(e, &e->cur, coef, n, comp, coef_scan_idx(m, zc, n, comp), m->ts[comp][zc]);
}
/* xEncIntraHeader, TEncSearch.cpp:965-1032 (I slice, no PCM) */
static void code_skip_flag(Enc *e, Cabac *c, int z);
static void enc_intra_header(Enc *e, const TU *t, int bLuma, int bChroma)
{
  const CtuMeta *m = e->cm; const int relZ = t->relZ;
  if (bLuma) {
    if (relZ == 0 && e->is) { code_skip_flag(e, &e->cur, t->cuZ); enc_bin(&e->cur, C_PRED_MODE, 1); }   /* P/B slices: skip flag + pred mode, TEncSearch.cpp:975-984 */
    if (relZ == 0 && t->cuDepth == 3) enc_bin(&e->cur, C_PART, m->part[t->cuZ] == SIZE_2Nx2N);   /* codePartSize, TEncSbac.cpp:431 */
    if (m->part[t->cuZ] == SIZE_2Nx2N) { if (relZ == 0) code_intra_dir_luma(e, &e->cur, t->cuZ, 0); }
    else { const int q = t->cuParts >> 2; if (t->trDepth > 0 && (relZ % q) == 0) code_intra_dir_luma(e, &e->cur, t->cuZ + relZ, 0); }
  }
  if (bChroma && relZ == 0) code_intra_dir_chroma(e, &e->cur, t->cuZ + relZ);
}
/* xGetIntraBitsQT, TEncSearch.cpp:1038-1060 */
static uint32_t intra_bits_qt(Enc *e, const TU *t, int bLuma, int bChroma)
{
  reset_bits(&e->cur);
  enc_intra_header(e, t, bLuma, bChroma);
  enc_subdiv_cbf_qt(e, t, bLuma, bChroma);
  if (bLuma) enc_coeff_qt(e, t, 0);
  if (bChroma) { enc_coeff_qt(e, t, 1); enc_coeff_qt(e, t, 2); }
  return num_bits(&e->cur);
}

/* ============================================================================================ */
/* one TU: predict, transform, RDOQ, reconstruct (TEncSearch::xIntraCodingTUBlock :1074-1357)    */
/* ============================================================================================ */
/* save1load2: 0 default, 1 save the prediction, 2 reuse the saved prediction */
static void intra_coding_tu_block(Enc *e, const TU *t, int comp, uint32_t *dist, int save1load2)
{
  CtuMeta *m = e->cm;
  if (comp && !t->cW) return;
  const int n = comp ? t->cW : (1 << t->log2);
  const int relZ = comp ? t->cRelZ : t->relZ;       /* GetAbsPartIdxTU() of the (iterator) TU */
  const int z = t->cuZ + relZ;
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;          /* position in the CTU, component samples */
  const int st = comp ? 32 : 64;
  const int layer = 5 - t->log2;
  const int parts = comp ? t->cParts : t->parts;
  Pel *org = e->org[comp] + (e->ctuY * st + by) * e->stride[comp] + e->ctuX * st + bx;
  Pel *pred = e->pred[comp] + by * st + bx, *resi = e->resi[comp] + by * st + bx;
  Pel *recQt = e->qtRec[layer][comp] + by * st + bx;
  Pel *recPic = e->rec[comp] + (e->ctuY * st + by) * e->stride[comp] + e->ctuX * st + bx;
  TCoeff *coef = e->qtCoef[comp][layer] + (comp ? t->cOff : z * 16);
  const int tskip = m->ts[comp][z];
  int mode = comp ? m->dirC[z] : m->dirL[z];
  if (comp && mode == DM_CHROMA_IDX) mode = m->dirL[z & ~3];
  if (save1load2 != 2) {
    const int filt = use_filtered_refs(comp, mode, n);
    const int x4 = e->ctuX * 16 + (Z2R[z] & 15), y4 = e->ctuY * 16 + (Z2R[z] >> 4);
    init_adi_pattern(e, comp, e->ctuX * st + bx, e->ctuY * st + by, n, x4, y4, comp ? n / 2 : n / 4, filt);
    pred_intra(e, comp, mode, n, filt, pred, st);
    if (save1load2 == 1) for (int y = 0; y < n; y++) memcpy(e->tsPred[comp] + y * n, pred + y * st, sizeof(Pel) * n);
  } else for (int y = 0; y < n; y++) memcpy(pred + y * st, e->tsPred[comp] + y * n, sizeof(Pel) * n);
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) resi[y * st + x] = (Pel)(org[y * e->stride[comp] + x] - pred[y * st + x]);
  if (comp == 0) memset(m->tr + z, t->trDepth, parts);      /* setTrIdxSubParts, TEncSearch.cpp:1229 */
  /* TComTrQuant::transformNxN, TComTrQuant.cpp:1337-1421 */
  TCoeff blk[1024], tc[1024];
  const int bd = e->cfg.bit_depth, log2n = ilog2(n);
  if (tskip) { const int sh = 15 - bd - log2n; for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) tc[y * n + x] = (TCoeff)resi[y * st + x] << sh; }
  else {
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) blk[y * n + x] = resi[y * st + x];
    hmo_fwd_transform(bd, blk, tc, n, comp == 0 && n == 4);
  }
  const int cbfCtx = comp ? 5 + t->trDepth : (t->trDepth == 0 ? 1 : 0);
  const int absSum = rdoq(e, tc, coef, n, comp, coef_scan_idx(m, z, n, comp), tskip, cbfCtx);
  memset(m->cbf[comp] + z, (absSum > 0 ? 1 : 0) << t->trDepth, parts);   /* setCbfPartRange, TComTrQuant.cpp:1419 */
  if (absSum > 0) { /* invTransformNxN, TComTrQuant.cpp:1423-1545 */
    dequant(e, coef, tc, n, comp);
    if (tskip) { const int sh = 15 - bd - log2n; const int off = sh == 0 ? 0 : (1 << (sh - 1)); for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) resi[y * st + x] = (Pel)((tc[y * n + x] + off) >> sh); }
    else { hmo_inv_transform(bd, tc, blk, n, comp == 0 && n == 4); for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) resi[y * st + x] = (Pel)blk[y * n + x]; }
  } else {
    memset(coef, 0, sizeof(TCoeff) * n * n);
    for (int y = 0; y < n; y++) memset(resi + y * st, 0, sizeof(Pel) * n);
  }
  const int maxv = (1 << bd) - 1;
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
    const Pel r = (Pel)clip3(0, maxv, pred[y * st + x] + resi[y * st + x]);
    pred[y * st + x] = r; recQt[y * st + x] = r; recPic[y * e->stride[comp] + x] = r;     /* piReco aliases piPred */
  }
  uint32_t d = hmo_sse(org, e->stride[comp], pred, st, n, n, bd);
  if (comp) d = (uint32_t)(e->chromaWeight * (double)d);                                  /* getDistPart, TComRdCost.cpp:447-450 */
  *dist += d;
}

/* xStoreIntraResultQT / xLoadIntraResultQT, TEncSearch.cpp:1790-1880 (one component) */
static void store_intra_result_qt(Enc *e, const TU *t, int comp)
{
  const int n = comp ? t->cW : (1 << t->log2), st = comp ? 32 : 64, layer = 5 - t->log2;
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
  memcpy(e->tsCoef[comp], e->qtCoef[comp][layer] + (comp ? t->cOff : (t->cuZ + t->relZ) * 16), sizeof(TCoeff) * n * n);
  for (int y = 0; y < n; y++) memcpy(e->tsRec[comp] + y * n, e->qtRec[layer][comp] + (by + y) * st + bx, sizeof(Pel) * n);
}
static void load_intra_result_qt(Enc *e, const TU *t, int comp)
{
  const int n = comp ? t->cW : (1 << t->log2), st = comp ? 32 : 64, layer = 5 - t->log2;
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
  memcpy(e->qtCoef[comp][layer] + (comp ? t->cOff : (t->cuZ + t->relZ) * 16), e->tsCoef[comp], sizeof(TCoeff) * n * n);
  Pel *recPic = e->rec[comp] + (e->ctuY * st + by) * e->stride[comp] + e->ctuX * st + bx;
  for (int y = 0; y < n; y++) {
    memcpy(e->qtRec[layer][comp] + (by + y) * st + bx, e->tsRec[comp] + y * n, sizeof(Pel) * n);
    memcpy(recPic + y * e->stride[comp], e->tsRec[comp] + y * n, sizeof(Pel) * n);
  }
}

/* ============================================================================================ */
/* luma residual quadtree (TEncSearch::xRecurIntraCodingQT :1364-1733, bLumaOnly = true)         */
/* ============================================================================================ */
static void recur_intra_coding_qt(Enc *e, const TU *t, uint32_t *distY, int checkFirst, double *rdCost)
{
  CtuMeta *m = e->cm;
  const int z = t->cuZ + t->relZ, fullDepth = t->cuDepth + t->trDepth, log2 = t->log2;
  const int nxn = m->part[t->cuZ] == SIZE_NxN;
  int checkFull = log2 <= 5;
  int checkSplit = log2 > tr_min_size_in_cu(6 - t->cuDepth, nxn);
  if (checkFirst && checkFull) checkSplit = 0;                            /* HHI_RQT_INTRA_SPEEDUP */
  double singleCost = MAX_DOUBLE; uint32_t singleDist = 0, singleCbf = 0; int bestModeId = 0;
  const int checkTS = (log2 == 2) && (m->part[z] == SIZE_NxN);           /* TransformSkip + TransformSkipFast */
  if (checkFull) {
    if (checkTS) {
      e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->cur;
      for (int modeId = 0; modeId < 2; modeId++) {
        uint32_t distTmp = 0; double costTmp;
        memset(m->ts[0] + z, modeId, t->parts);
        intra_coding_tu_block(e, t, 0, &distTmp, modeId == 0 ? 1 : 2);
        const uint32_t cbfTmp = (m->cbf[0][z] >> t->trDepth) & 1;
        if (modeId == 1 && cbfTmp == 0) costTmp = MAX_DOUBLE;
        else { const uint32_t bits = intra_bits_qt(e, t, 1, 0); costTmp = calc_rd_cost(e, bits, distTmp); }
        if (costTmp < singleCost) {
          singleCost = costTmp; singleDist = distTmp; singleCbf = cbfTmp; bestModeId = modeId;
          if (bestModeId == 0) { store_intra_result_qt(e, t, 0); e->slot[fullDepth][CI_TEMP_BEST] = e->cur; }
        }
        if (modeId == 0) e->cur = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
      }
      memset(m->ts[0] + z, bestModeId, t->parts);
      if (bestModeId == 0) {
        load_intra_result_qt(e, t, 0);
        memset(m->cbf[0] + z, singleCbf << t->trDepth, t->parts);
        e->cur = e->slot[fullDepth][CI_TEMP_BEST];
      }
    } else {
      if (checkSplit) e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->cur;
      memset(m->ts[0] + z, 0, t->parts);
      intra_coding_tu_block(e, t, 0, &singleDist, 0);
      if (checkSplit) singleCbf = (m->cbf[0][z] >> t->trDepth) & 1;
      const uint32_t bits = intra_bits_qt(e, t, 1, 0);
      singleCost = calc_rd_cost(e, bits, singleDist);
    }
  }
  if (checkSplit) {
    if (checkFull) { e->slot[fullDepth][CI_QT_TRAFO_TEST] = e->cur; e->cur = e->slot[fullDepth][CI_QT_TRAFO_ROOT]; }
    else e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->cur;
    double splitCost = 0.0; uint32_t splitDist = 0, splitCbf = 0;
    for (int s = 0; s < 4; s++) {
      TU ch = tu_child(t, s, 0);
      recur_intra_coding_qt(e, &ch, &splitDist, checkFirst, &splitCost);
      splitCbf |= (m->cbf[0][ch.cuZ + ch.relZ] >> ch.trDepth) & 1;
    }
    if (splitCbf) for (int o = 0; o < t->parts; o++) m->cbf[0][z + o] |= (uint8_t)(1 << t->trDepth);
    e->cur = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
    const uint32_t splitBits = intra_bits_qt(e, t, 1, 0);
    splitCost = calc_rd_cost(e, splitBits, splitDist);
    if (splitCost < singleCost) { *distY += splitDist; *rdCost += splitCost; return; }
    e->cur = e->slot[fullDepth][CI_QT_TRAFO_TEST];
    memset(m->tr + z, t->trDepth, t->parts);
    memset(m->cbf[0] + z, singleCbf << t->trDepth, t->parts);
    memset(m->ts[0] + z, bestModeId, t->parts);
    { /* reconstruction of the unsplit TU back into the picture for the following blocks */
      const int n = 1 << log2, layer = 5 - log2;
      Pel *recPic = e->rec[0] + (e->ctuY * 64 + t->y) * e->stride[0] + e->ctuX * 64 + t->x;
      for (int y = 0; y < n; y++) memcpy(recPic + y * e->stride[0], e->qtRec[layer][0] + (t->y + y) * 64 + t->x, sizeof(Pel) * n);
    }
  }
  *distY += singleDist; *rdCost += singleCost;
}

/* xSetIntraResultQT, TEncSearch.cpp:1737-1788 (luma only) */
static void set_intra_result_qt(Enc *e, const TU *t)
{
  const CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  if (m->tr[z] == t->trDepth) {
    const int n = 1 << t->log2, layer = 5 - t->log2;
    memcpy(e->cc[0] + z * 16, e->qtCoef[0][layer] + z * 16, sizeof(TCoeff) * n * n);
    for (int y = 0; y < n; y++) memcpy(e->reco[0] + (t->y + y) * 64 + t->x, e->qtRec[layer][0] + (t->y + y) * 64 + t->x, sizeof(Pel) * n);
  } else for (int s = 0; s < 4; s++) { TU ch = tu_child(t, s, 0); set_intra_result_qt(e, &ch); }
}

/* ============================================================================================ */
/* luma mode decision of one CU (TEncSearch::estIntraPredQT :2289-2692)                          */
/* ============================================================================================ */
static uint32_t est_intra_pred_qt(Enc *e, int cuZ, int cuDepth)
{
  CtuMeta *m = e->cm;
  const int cuParts = 256 >> (2 * cuDepth);
  const int nxn = m->part[cuZ] == SIZE_NxN;
  const int numPU = nxn ? 4 : 1, puParts = cuParts / numPU;
  const int cuLog2 = 6 - cuDepth, puLog2 = cuLog2 - nxn, n = 1 << puLog2;
  const int bd = e->cfg.bit_depth;
  uint32_t overallDistY = 0;
  TU root = tu_root(cuZ, cuDepth);
  for (int pu = 0; pu < numPU; pu++) {
    TU t = nxn ? tu_child(&root, pu, 0) : root;
    const int z = cuZ + t.relZ;
    int numModesForFullRD = INTRA_MODE_NUM_FAST[puLog2 - 1];
    int rdModeList[35]; double candCost[35];
    { /* SATD pre-selection over the 35 modes, :2360-2410 */
      const int x4 = e->ctuX * 16 + (Z2R[z] & 15), y4 = e->ctuY * 16 + (Z2R[z] >> 4);
      init_adi_pattern(e, 0, e->ctuX * 64 + t.x, e->ctuY * 64 + t.y, n, x4, y4, n / 4, 1);
      for (int i = 0; i < numModesForFullRD; i++) candCost[i] = MAX_DOUBLE;
      Pel *org = e->org[0] + (e->ctuY * 64 + t.y) * e->stride[0] + e->ctuX * 64 + t.x;
      Pel *pred = e->pred[0] + t.y * 64 + t.x;
      for (int mode = 0; mode < 35; mode++) {
        pred_intra(e, 0, mode, n, use_filtered_refs(0, mode, n), pred, 64);
        const uint32_t sad = hmo_hads(org, e->stride[0], pred, 64, n, n, bd);
        /* xModeBitsIntra, TEncSearch.cpp:5456-5478 */
        e->cur.frac = e->slot[cuDepth][CI_CURR_BEST].frac; e->cur.s[C_INTRA_LUMA] = e->slot[cuDepth][CI_CURR_BEST].s[C_INTRA_LUMA];
        const uint8_t orig = m->dirL[z]; m->dirL[z] = (uint8_t)mode;
        reset_bits(&e->cur); code_intra_dir_luma(e, &e->cur, z, 0);
        m->dirL[z] = orig;
        const uint32_t modeBits = num_bits(&e->cur);
        const double cost = (double)sad + (double)modeBits * e->sqrtLambda;
        /* xUpdateCandList, TEncSearch.cpp:5484-5505 */
        int shift = 0;
        while (shift < numModesForFullRD && cost < candCost[numModesForFullRD - 1 - shift]) shift++;
        if (shift != 0) {
          for (int i = 1; i < shift; i++) { rdModeList[numModesForFullRD - i] = rdModeList[numModesForFullRD - 1 - i]; candCost[numModesForFullRD - i] = candCost[numModesForFullRD - 1 - i]; }
          rdModeList[numModesForFullRD - shift] = mode; candCost[numModesForFullRD - shift] = cost;
        }
      }
      int preds[3], iMode = -1;
      int numCand = intra_dir_predictor(e, z, preds, &iMode);
      if (iMode >= 0) numCand = iMode;
      for (int j = 0; j < numCand; j++) {
        int included = 0;
        for (int i = 0; i < numModesForFullRD; i++) included |= (preds[j] == rdModeList[i]);
        if (!included) rdModeList[numModesForFullRD++] = preds[j];
      }
    }
    int bestPUMode = 0; uint32_t bestPUDistY = 0; double bestPUCost = MAX_DOUBLE;
    for (int pass = 0; pass <= numModesForFullRD; pass++) {
      /* passes 0..N-1: candidates with the unsplit TU (bCheckFirst); last pass: best mode with the full RQT */
      const int last = (pass == numModesForFullRD);
      const int orgMode = last ? bestPUMode : rdModeList[pass];
      memset(m->dirL + z, orgMode, puParts);
      e->cur = e->slot[cuDepth][CI_CURR_BEST];
      uint32_t puDistY = 0; double puCost = 0.0;
      recur_intra_coding_qt(e, &t, &puDistY, !last, &puCost);
      if (puCost < bestPUCost) {
        bestPUMode = orgMode; bestPUDistY = puDistY; bestPUCost = puCost;
        set_intra_result_qt(e, &t);
        memcpy(e->tmpTr, m->tr + z, puParts);
        for (int c = 0; c < 3; c++) { memcpy(e->tmpCbf[c], m->cbf[c] + z, puParts); memcpy(e->tmpTs[c], m->ts[c] + z, puParts); }
      }
    }
    overallDistY += bestPUDistY;
    memcpy(m->tr + z, e->tmpTr, puParts);
    for (int c = 0; c < 3; c++) { memcpy(m->cbf[c] + z, e->tmpCbf[c], puParts); memcpy(m->ts[c] + z, e->tmpTs[c], puParts); }
    if (pu != numPU - 1) { /* reconstruction for the next PU, :2632-2660 */
      Pel *recPic = e->rec[0] + (e->ctuY * 64 + t.y) * e->stride[0] + e->ctuX * 64 + t.x;
      for (int y = 0; y < n; y++) memcpy(recPic + y * e->stride[0], e->reco[0] + (t.y + y) * 64 + t.x, sizeof(Pel) * n);
    }
    memset(m->dirL + z, bestPUMode, puParts);
  }
  if (numPU > 1) {
    uint8_t comb[3] = {0, 0, 0};
    for (int p = 0; p < 4; p++) for (int c = 0; c < 3; c++) comb[c] |= (m->cbf[c][cuZ + p * puParts] >> 1) & 1;
    for (int o = 0; o < cuParts; o++) for (int c = 0; c < 3; c++) m->cbf[c][cuZ + o] |= comb[c];
  }
  e->cur = e->slot[cuDepth][CI_CURR_BEST];
  return overallDistY;
}

/* ============================================================================================ */
/* chroma (TEncSearch::xRecurIntraChromaCodingQT :1958-2145, estIntraPredChromaQT :2698-2849)    */
/* ============================================================================================ */
static void recur_intra_chroma_coding_qt(Enc *e, const TU *t, uint32_t *dist)
{
  CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  if (m->tr[z] == t->trDepth) {
    if (!t->cW) return;
    const int fullDepth = t->cuDepth + t->trDepth;
    int checkTS = (t->cW == 4) && (t->log2 == 2);
    if (checkTS) { int nb = 0; for (int s = 0; s < 4; s++) nb += m->ts[0][z + s]; checkTS = nb > 0; }
    const int zc = t->cuZ + t->cRelZ;
    for (int comp = 1; comp < 3; comp++) {
      e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->cur;
      double singleCost = MAX_DOUBLE; uint32_t singleDistC = 0, singleCbfC = 0; int bestTS = 0, bestModeId = 0, currModeId = 0;
      double costTmp = 0; const int total = checkTS ? 2 : 1;
      for (int tsMode = 0; tsMode < total; tsMode++) {
        memset(m->ts[comp] + zc, tsMode, t->cParts);
        currModeId++;
        const int isOne = (total == 1), isLast = (currModeId == total);
        const int sl = isOne ? 0 : (tsMode == 0 ? 1 : 2);
        uint32_t distTmp = 0;
        intra_coding_tu_block(e, t, comp, &distTmp, sl);
        const uint32_t cbfTmp = (m->cbf[comp][zc] >> t->trDepth) & 1;
        if (tsMode == 1 && cbfTmp == 0) costTmp = MAX_DOUBLE;
        else if (!isOne) { /* xGetIntraBitsQTChroma, TEncSearch.cpp:1062-1070 */
          reset_bits(&e->cur); enc_coeff_qt(e, t, comp); costTmp = calc_rd_cost(e, num_bits(&e->cur), distTmp);
        }
        if (costTmp < singleCost) {
          singleCost = costTmp; singleDistC = distTmp; bestTS = tsMode; bestModeId = currModeId; singleCbfC = cbfTmp;
          if (!isOne && !isLast) { store_intra_result_qt(e, t, comp); e->slot[fullDepth][CI_TEMP_BEST] = e->cur; }
        }
        if (!isOne && !isLast) e->cur = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
      }
      if (bestModeId < total) {
        load_intra_result_qt(e, t, comp);
        memset(m->cbf[comp] + zc, singleCbfC << t->trDepth, t->cParts);
        e->cur = e->slot[fullDepth][CI_TEMP_BEST];
      }
      memset(m->ts[comp] + zc, bestTS, t->cParts);
      *dist += singleDistC;
    }
  } else {
    uint32_t splitCbf[3] = {0, 0, 0};
    for (int s = 0; s < 4; s++) {
      TU ch = tu_child(t, s, 0);
      recur_intra_chroma_coding_qt(e, &ch, dist);
      for (int c = 1; c < 3; c++) splitCbf[c] |= (m->cbf[c][ch.cuZ + ch.relZ] >> ch.trDepth) & 1;
    }
    for (int c = 1; c < 3; c++) if (splitCbf[c]) for (int o = 0; o < t->parts; o++) m->cbf[c][z + o] |= (uint8_t)(1 << t->trDepth);
  }
}
/* xSetIntraResultChromaQT, TEncSearch.cpp:2150-2200 */
static void set_intra_result_chroma_qt(Enc *e, const TU *t)
{
  const CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  if (!t->cW) return;
  if (m->tr[z] == t->trDepth) {
    const int n = t->cW, layer = 5 - t->log2;
    for (int c = 1; c < 3; c++) {
      memcpy(e->cc[c] + t->cOff, e->qtCoef[c][layer] + t->cOff, sizeof(TCoeff) * n * n);
      for (int y = 0; y < n; y++) memcpy(e->reco[c] + (t->cy + y) * 32 + t->cx, e->qtRec[layer][c] + (t->cy + y) * 32 + t->cx, sizeof(Pel) * n);
    }
  } else for (int s = 0; s < 4; s++) { TU ch = tu_child(t, s, 0); set_intra_result_chroma_qt(e, &ch); }
}
static uint32_t est_intra_pred_chroma_qt(Enc *e, int cuZ, int cuDepth)
{
  CtuMeta *m = e->cm; const int cuParts = 256 >> (2 * cuDepth);
  TU t = tu_root(cuZ, cuDepth);
  int bestMode = 0; uint32_t bestDist = 0; double bestCost = MAX_DOUBLE;
  int modeList[5] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX, DM_CHROMA_IDX};       /* getAllowedChromaDir, TComDataCU.cpp:1486 */
  for (int i = 0; i < 4; i++) if (m->dirL[cuZ] == modeList[i]) { modeList[i] = 34; break; }
  static uint8_t saveCbf[3][256], saveTs[3][256];
  for (int mi = 0; mi < 5; mi++) {
    e->cur = e->slot[cuDepth][CI_CURR_BEST];
    uint32_t dist = 0;
    memset(m->dirC + cuZ, modeList[mi], cuParts);
    recur_intra_chroma_coding_qt(e, &t, &dist);
    e->cur = e->slot[cuDepth][CI_CURR_BEST];
    const uint32_t bits = intra_bits_qt(e, &t, 0, 1);
    const double cost = calc_rd_cost(e, bits, dist);
    if (cost < bestCost) {
      bestCost = cost; bestDist = dist; bestMode = modeList[mi];
      set_intra_result_chroma_qt(e, &t);
      for (int c = 1; c < 3; c++) { memcpy(saveCbf[c], m->cbf[c] + cuZ, cuParts); memcpy(saveTs[c], m->ts[c] + cuZ, cuParts); }
    }
  }
  for (int c = 1; c < 3; c++) { memcpy(m->cbf[c] + cuZ, saveCbf[c], cuParts); memcpy(m->ts[c] + cuZ, saveTs[c], cuParts); }
  memset(m->dirC + cuZ, bestMode, cuParts);
  e->cur = e->slot[cuDepth][CI_CURR_BEST];
  return bestDist;
}

/* ============================================================================================ */
/* final syntax of a CU (TEncEntropy::xEncodeTransform, TEncEntropy.cpp:222-412)                 */
/* ============================================================================================ */
static void code_delta_qp(Enc *e, Cabac *c, int dqp)
{ /* TEncSbac::codeDeltaQP, TEncSbac.cpp:870-895 */
  const int off = 6 * (e->cfg.bit_depth - 8);
  dqp = (dqp + 78 + off + (off / 2)) % (52 + off) - 26 - (off / 2);
  const unsigned a = (unsigned)(dqp > 0 ? dqp : -dqp), tu = a < 5 ? a : 5;
  enc_bin(c, C_DQP, tu ? 1 : 0);                               /* xWriteUnaryMaxSymbol(tu, ctx, 1, CU_DQP_TU_CMAX), :281 */
  if (tu) { unsigned k = tu; while (--k) enc_bin(c, C_DQP + 1, 1); if (5 > tu) enc_bin(c, C_DQP + 1, 0); }
  if (a >= 5) {                                                /* xWriteEpExGolomb(a - 5, CU_DQP_EG_k = 0), :309 */
    unsigned sym = a - 5, count = 0, bins = 0; int nb = 0;
    while (sym >= (1u << count)) { bins = 2 * bins + 1; nb++; sym -= 1u << count; count++; }
    bins = 2 * bins; nb++;
    bins = (bins << count) | sym; nb += (int)count;
    enc_epv(c, bins, nb);
  }
  if (a > 0) enc_epv(c, dqp > 0 ? 0 : 1, 1);
}
static void encode_transform(Enc *e, Cabac *c, const TU *t, int *codeDqp)
{
  const CtuMeta *m = e->cm; const int z = t->cuZ + t->relZ;
  const int subdiv = m->tr[z] > t->trDepth;
  const int nxn = m->part[z] == SIZE_NxN;
  uint32_t cbf[3]; int any = 0;
  for (int comp = 0; comp < 3; comp++) { cbf[comp] = (m->cbf[comp][z] >> t->trDepth) & 1; any |= cbf[comp] != 0; }
  if (nxn && t->trDepth == 0) { }
  else if (t->log2 > 5) { }
  else if (t->log2 == 2) { }
  else if (t->log2 == tr_min_size_in_cu(6 - t->cuDepth, nxn)) { }
  else enc_bin(c, C_SUBDIV + (5 - t->log2), subdiv);
  const int first = t->trDepth == 0;
  for (int comp = 1; comp < 3; comp++)
    if (first || t->cCodeAll)
      if (first || ((m->cbf[comp][z] >> (t->trDepth - 1)) & 1)) code_qt_cbf(e, c, t, comp, subdiv == 0);
  if (subdiv) { for (int s = 0; s < 4; s++) { TU ch = tu_child(t, s, 1); encode_transform(e, c, &ch, codeDqp); } return; }
  code_qt_cbf(e, c, t, 0, 1);
  if (!any) return;
  if (e->dq && codeDqp && *codeDqp) { code_delta_qp(e, c, m->qp[t->cuZ] - e->refQp); *codeDqp = 0; }   /* "dQP: only for CTU once", TEncEntropy.cpp:343-351 */
  for (int comp = 0; comp < 3; comp++) {
    if (comp && !t->cW) continue;
    if (!cbf[comp]) continue;
    const int n = comp ? t->cW : (1 << t->log2);
    const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
    const TCoeff *coef = e->cc[comp] + (comp ? t->cOff : z * 16);
    code_coeff_nxn(e, c, coef, n, comp, coef_scan_idx(m, zc, n, comp), m->ts[comp][zc]);
  }
}
/* the CU-level syntax shared by xCheckRDCostIntra (:1601-1626) and xEncodeCU (:1246-1288), I slice */
static void encode_cu_syntax(Enc *e, Cabac *c, int cuZ, int cuDepth)
{ /* the delta QP goes with the first coded block while TEncCu::m_bEncodeDQP is set -- in the RD search too (xCheckRDCostIntra :1629-1633 hands the
     member through as it stands, i.e. as the previous CTU's encodeCtu left it) */
  const CtuMeta *m = e->cm;
  if (e->is) { code_skip_flag(e, c, cuZ); enc_bin(c, C_PRED_MODE, 1); }
  if (cuDepth == 3) enc_bin(c, C_PART, m->part[cuZ] == SIZE_2Nx2N);
  code_intra_dir_luma(e, c, cuZ, 1);
  code_intra_dir_chroma(e, c, cuZ);
  TU t = tu_root(cuZ, cuDepth);
  encode_transform(e, c, &t, &e->dqpFlag);
}

/* ============================================================================================ */
/* CU quadtree (TEncCu::xCompressCU :466-1122, xCheckRDCostIntra :1574-1646)                     */
/* ============================================================================================ */
static void init_est_data(Enc *e, int cuZ, int cuDepth)
{ /* TComDataCU::initEstData, TComDataCU.cpp:484-552 */
  CtuMeta *m = e->cm; const int parts = 256 >> (2 * cuDepth);
  memset(m->depth + cuZ, cuDepth, parts); memset(m->part + cuZ, SIZE_NONE, parts); memset(m->pred + cuZ, MODE_NONE, parts);
  memset(m->dirL + cuZ, DC_IDX, parts); memset(m->dirC + cuZ, 0, parts); memset(m->tr + cuZ, 0, parts);
  for (int c = 0; c < 3; c++) { memset(m->cbf[c] + cuZ, 0, parts); memset(m->ts[c] + cuZ, 0, parts); }
  memset(m->skip + cuZ, 0, parts); memset(m->mrg + cuZ, 0, parts); memset(m->mrgIdx + cuZ, 0, parts); memset(m->interDir + cuZ, 0, parts);
  memset(m->qp + cuZ, e->ctuQp, parts);
  for (int l = 0; l < 2; l++) {
    memset(m->mv[l] + cuZ, 0, sizeof(Mv) * parts); memset(m->mvd[l] + cuZ, 0, sizeof(Mv) * parts);
    memset(m->refIdx[l] + cuZ, -1, parts); memset(m->mvpIdx[l] + cuZ, -1, parts); memset(m->mvpNum[l] + cuZ, -1, parts);
  }
  memset(e->cc[0] + cuZ * 16, 0, sizeof(TCoeff) * parts * 16);
  memset(e->cc[1] + cuZ * 4, 0, sizeof(TCoeff) * parts * 4);
  memset(e->cc[2] + cuZ * 4, 0, sizeof(TCoeff) * parts * 4);
}
static void copy_cu_planes(Pel *const dst[3], const int dstStride[3], Pel *const src[3], const int srcStride[3], int x, int y, int n)
{
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, nn = n >> sh;
    for (int r = 0; r < nn; r++) memcpy(dst[c] + ((y >> sh) + r) * dstStride[c] + (x >> sh), src[c] + ((y >> sh) + r) * srcStride[c] + (x >> sh), sizeof(Pel) * nn);
  }
}
static void meta_copy_inter(CtuMeta *d, const CtuMeta *s, int z, int parts)
{
  memcpy(d->skip + z, s->skip + z, parts); memcpy(d->mrg + z, s->mrg + z, parts); memcpy(d->mrgIdx + z, s->mrgIdx + z, parts); memcpy(d->interDir + z, s->interDir + z, parts);
  for (int l = 0; l < 2; l++) {
    memcpy(d->mv[l] + z, s->mv[l] + z, sizeof(Mv) * parts); memcpy(d->mvd[l] + z, s->mvd[l] + z, sizeof(Mv) * parts);
    memcpy(d->refIdx[l] + z, s->refIdx[l] + z, parts); memcpy(d->mvpIdx[l] + z, s->mvpIdx[l] + z, parts); memcpy(d->mvpNum[l] + z, s->mvpNum[l] + z, parts);
  }
  memcpy(d->qp + z, s->qp + z, parts);
}
static void save_best(Enc *e, int cuZ, int cuDepth, double cost, uint32_t bits, uint32_t dist)
{
  Best *b = &e->best[cuDepth]; const CtuMeta *m = e->cm; const int parts = 256 >> (2 * cuDepth);
  const uint8_t *src[12] = {m->depth, m->part, m->pred, m->dirL, m->dirC, m->tr, m->cbf[0], m->cbf[1], m->cbf[2], m->ts[0], m->ts[1], m->ts[2]};
  uint8_t *dst[12] = {b->m.depth, b->m.part, b->m.pred, b->m.dirL, b->m.dirC, b->m.tr, b->m.cbf[0], b->m.cbf[1], b->m.cbf[2], b->m.ts[0], b->m.ts[1], b->m.ts[2]};
  for (int i = 0; i < 12; i++) memcpy(dst[i] + cuZ, src[i] + cuZ, parts);
  meta_copy_inter(&b->m, m, cuZ, parts);
  memcpy(b->coef[0] + cuZ * 16, e->cc[0] + cuZ * 16, sizeof(TCoeff) * parts * 16);
  memcpy(b->coef[1] + cuZ * 4, e->cc[1] + cuZ * 4, sizeof(TCoeff) * parts * 4);
  memcpy(b->coef[2] + cuZ * 4, e->cc[2] + cuZ * 4, sizeof(TCoeff) * parts * 4);
  Pel *d[3] = {b->reco[0], b->reco[1], b->reco[2]}, *s[3] = {e->reco[0], e->reco[1], e->reco[2]};
  const int st[3] = {64, 32, 32};
  copy_cu_planes(d, st, s, st, (Z2R[cuZ] & 15) * 4, (Z2R[cuZ] >> 4) * 4, 64 >> cuDepth);
  b->cost = cost; b->bits = bits; b->dist = dist;
}
static void restore_best(Enc *e, int cuZ, int cuDepth)
{ /* TComDataCU::copyToPic + TEncCu::xCopyYuv2Pic of the unsplit winner */
  Best *b = &e->best[cuDepth]; CtuMeta *m = e->cm; const int parts = 256 >> (2 * cuDepth);
  uint8_t *dst[12] = {m->depth, m->part, m->pred, m->dirL, m->dirC, m->tr, m->cbf[0], m->cbf[1], m->cbf[2], m->ts[0], m->ts[1], m->ts[2]};
  const uint8_t *src[12] = {b->m.depth, b->m.part, b->m.pred, b->m.dirL, b->m.dirC, b->m.tr, b->m.cbf[0], b->m.cbf[1], b->m.cbf[2], b->m.ts[0], b->m.ts[1], b->m.ts[2]};
  for (int i = 0; i < 12; i++) memcpy(dst[i] + cuZ, src[i] + cuZ, parts);
  meta_copy_inter(m, &b->m, cuZ, parts);
  memcpy(e->cc[0] + cuZ * 16, b->coef[0] + cuZ * 16, sizeof(TCoeff) * parts * 16);
  memcpy(e->cc[1] + cuZ * 4, b->coef[1] + cuZ * 4, sizeof(TCoeff) * parts * 4);
  memcpy(e->cc[2] + cuZ * 4, b->coef[2] + cuZ * 4, sizeof(TCoeff) * parts * 4);
  const int x = (Z2R[cuZ] & 15) * 4, y = (Z2R[cuZ] >> 4) * 4, n = 64 >> cuDepth;
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, nn = n >> sh, st = c ? 32 : 64;
    Pel *recPic = e->rec[c] + (e->ctuY * st + (y >> sh)) * e->stride[c] + e->ctuX * st + (x >> sh);
    for (int r = 0; r < nn; r++) memcpy(recPic + r * e->stride[c], b->reco[c] + ((y >> sh) + r) * st + (x >> sh), sizeof(Pel) * nn);
  }
}

/* xCheckRDCostIntra; returns through cost / bits / dist, leaves the trial in place */
/* TEncCu::xCheckDQP :1742-1763 (RDO_WITHOUT_DQP_BITS 0): a candidate of quantisation-group size or larger -- with MaxCuDQPDepth 0 the 64x64 CU --
   pays for its delta QP when it has a coded block, and falls back to the predicted QP when it has none */
static void check_dqp(Enc *e, int cuZ, int cuDepth, double *cost, uint32_t *bits, uint32_t dist)
{
  if (!e->dq || cuDepth != 0) return;
  CtuMeta *m = e->cm;
  if ((m->cbf[0][cuZ] & 1) || (m->cbf[1][cuZ] & 1) || (m->cbf[2][cuZ] & 1)) {     /* getQtRootCbf(0) */
    reset_bits(&e->cur);
    code_delta_qp(e, &e->cur, m->qp[cuZ] - e->refQp);
    *bits += num_bits(&e->cur);
    *cost = calc_rd_cost(e, *bits, dist);
  } else memset(m->qp + cuZ, e->refQp, 256 >> (2 * cuDepth));
}
static void check_rd_cost_intra(Enc *e, int cuZ, int cuDepth, int partSize, double *cost, uint32_t *bits, uint32_t *dist)
{
  CtuMeta *m = e->cm; const int parts = 256 >> (2 * cuDepth);
  init_est_data(e, cuZ, cuDepth);
  memset(m->part + cuZ, partSize, parts); memset(m->pred + cuZ, MODE_INTRA, parts);
  uint32_t d = est_intra_pred_qt(e, cuZ, cuDepth);
  { /* luma reconstruction of the CU into the picture, TEncCu.cpp:1608 */
    const int x = (Z2R[cuZ] & 15) * 4, y = (Z2R[cuZ] >> 4) * 4, n = 64 >> cuDepth;
    Pel *recPic = e->rec[0] + (e->ctuY * 64 + y) * e->stride[0] + e->ctuX * 64 + x;
    for (int r = 0; r < n; r++) memcpy(recPic + r * e->stride[0], e->reco[0] + (y + r) * 64 + x, sizeof(Pel) * n);
  }
  d += est_intra_pred_chroma_qt(e, cuZ, cuDepth);
  reset_bits(&e->cur);
  encode_cu_syntax(e, &e->cur, cuZ, cuDepth);
  e->slot[cuDepth][CI_TEMP_BEST] = e->cur;
  *bits = num_bits(&e->cur); *dist = d;
  *cost = calc_rd_cost(e, *bits, *dist);
  check_dqp(e, cuZ, cuDepth, cost, bits, *dist);
}

#include "hm_oracle_inter.inc"

static void compress_cu(Enc *e, int cuZ, int cuDepth, int parentPartSize, double *outCost, uint32_t *outBits, uint32_t *outDist)
{
  CtuMeta *m = e->cm;
  const int size = 64 >> cuDepth, parts = 256 >> (2 * cuDepth);
  const int lx = e->ctuX * 64 + (Z2R[cuZ] & 15) * 4, ty = e->ctuY * 64 + (Z2R[cuZ] >> 4) * 4;
  const int boundary = !((lx + size - 1 < e->cfg.width) && (ty + size - 1 < e->cfg.height));
  double bestCost = MAX_DOUBLE; uint32_t bestBits = 0, bestDist = 0;
  if (!boundary && e->is) { /* P/B slice: TEncCu.cpp:628-836 */
    BestRd br = { MAX_DOUBLE, 0, 0 };
    compress_cu_inter_modes(e, cuZ, cuDepth, parentPartSize, &br);
    const CtuMeta *bm = &e->best[cuDepth].m;
    if (bm->cbf[0][cuZ] != 0 || bm->cbf[1][cuZ] != 0 || bm->cbf[2][cuZ] != 0) {   /* avoid very complex intra if it is unlikely, :820 */
      double c; uint32_t b, d;
      check_rd_cost_intra(e, cuZ, cuDepth, SIZE_2Nx2N, &c, &b, &d);
      check_best_mode(e, cuZ, cuDepth, &br, c, b, d);
      if (cuDepth == 3) { check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN, &c, &b, &d); check_best_mode(e, cuZ, cuDepth, &br, c, b, d); }
    }
    bestCost = br.cost; bestBits = br.bits; bestDist = br.dist;
    reset_bits(&e->cur);
    if (cuDepth != 3) enc_bin(&e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
    bestBits += num_bits(&e->cur);
    bestCost = calc_rd_cost(e, bestBits, bestDist);
    e->best[cuDepth].cost = bestCost; e->best[cuDepth].bits = bestBits;
  } else if (!boundary) {
    double c; uint32_t b, d;
    check_rd_cost_intra(e, cuZ, cuDepth, SIZE_2Nx2N, &c, &b, &d);
    if (c < bestCost) { bestCost = c; bestBits = b; bestDist = d; save_best(e, cuZ, cuDepth, c, b, d); e->slot[cuDepth][CI_NEXT_BEST] = e->slot[cuDepth][CI_TEMP_BEST]; }
    if (cuDepth == 3) {
      check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN, &c, &b, &d);
      if (c < bestCost) { bestCost = c; bestBits = b; bestDist = d; save_best(e, cuZ, cuDepth, c, b, d); e->slot[cuDepth][CI_NEXT_BEST] = e->slot[cuDepth][CI_TEMP_BEST]; }
    }
    /* split flag of the unsplit candidate, TEncCu.cpp:859-863: coded on the go-on coder in its current state.
       The context is derived from the BEST candidate's depth arrays and its neighbours. */
    reset_bits(&e->cur);
    if (cuDepth != 3) enc_bin(&e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
    bestBits += num_bits(&e->cur);
    bestCost = calc_rd_cost(e, bestBits, bestDist);
    e->best[cuDepth].cost = bestCost; e->best[cuDepth].bits = bestBits;
  }
  if (cuDepth < 3) {
    init_est_data(e, cuZ, cuDepth);
    double splitCost; uint32_t splitBits = 0, splitDist = 0;
    const int q = parts >> 2;
    for (int s = 0; s < 4; s++) {
      const int subZ = cuZ + s * q;
      const int sx = e->ctuX * 64 + (Z2R[subZ] & 15) * 4, sy = e->ctuY * 64 + (Z2R[subZ] >> 4) * 4;
      /* TComDataCU::initSubCU, TComDataCU.cpp:555-640 */
      memset(m->depth + subZ, cuDepth + 1, q); memset(m->part + subZ, SIZE_NONE, q); memset(m->pred + subZ, MODE_NONE, q);
      if (sx < e->cfg.width && sy < e->cfg.height) {
        if (s == 0) e->slot[cuDepth + 1][CI_CURR_BEST] = e->slot[cuDepth][CI_CURR_BEST];
        else e->slot[cuDepth + 1][CI_CURR_BEST] = e->slot[cuDepth + 1][CI_NEXT_BEST];
        double c; uint32_t b, d;
        const int bestIsInter = !boundary && e->best[cuDepth].m.pred[cuZ] == MODE_INTER;   /* rpcBestCU->isInter(0), :1026 */
        compress_cu(e, subZ, cuDepth + 1, bestIsInter ? e->best[cuDepth].m.part[cuZ] : SIZE_NONE, &c, &b, &d);
        splitBits += b; splitDist += d;
      }
    }
    if (!boundary) {
      reset_bits(&e->cur);
      enc_bin(&e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), m->depth[cuZ] > cuDepth);
      splitBits += num_bits(&e->cur);
    }
    splitCost = calc_rd_cost(e, splitBits, splitDist);
    if (e->dq && cuDepth == 0) { /* the split candidate of quantisation-group size, TEncCu.cpp:1052-1085 */
      int first = 256;             /* first CU (z order) with a coded block; setQPSubCUs (TComDataCU.cpp:1763-1787) gives every CU before it the predicted QP */
      for (int z = 0; z < 256; z++) if (m->cbf[0][z] || m->cbf[1][z] || m->cbf[2][z]) { first = z & ~((256 >> (2 * m->depth[z])) - 1); break; }
      if (first < 256) {
        reset_bits(&e->cur);
        code_delta_qp(e, &e->cur, m->qp[0] - e->refQp);
        splitBits += num_bits(&e->cur);
        splitCost = calc_rd_cost(e, splitBits, splitDist);
      }
      memset(m->qp, e->refQp, first);
    }
    e->slot[cuDepth][CI_TEMP_BEST] = e->slot[cuDepth + 1][CI_NEXT_BEST];
    if (splitCost < bestCost) {
      bestCost = splitCost; bestBits = splitBits; bestDist = splitDist;
      e->slot[cuDepth][CI_NEXT_BEST] = e->slot[cuDepth][CI_TEMP_BEST];
      /* the split configuration is already in place; refresh this depth's recon snapshot for the parent */
    } else restore_best(e, cuZ, cuDepth);
  } else restore_best(e, cuZ, cuDepth);
  *outCost = bestCost; *outBits = bestBits; *outDist = bestDist;
}

/* TEncCu::xEncodeCU, TEncCu.cpp:1185-1295: re-encode the decided CTU to advance the contexts */
static void encode_cu(Enc *e, Cabac *c, int z, int depth, int lastCtuOfSlice)
{
  const CtuMeta *m = e->cm; const int size = 64 >> depth;
  const int lx = e->ctuX * 64 + (Z2R[z] & 15) * 4, ty = e->ctuY * 64 + (Z2R[z] >> 4) * 4;
  const int inside = (lx + size - 1 < e->cfg.width) && (ty + size - 1 < e->cfg.height);
  if (inside && depth != 3) enc_bin(c, C_SPLIT + ctx_split_flag(e, z, depth), m->depth[z] > depth);
  if ((depth < m->depth[z] && depth < 3) || !inside) {
    const int q = (256 >> (2 * depth)) >> 2;
    for (int s = 0; s < 4; s++) {
      const int sz = z + s * q;
      const int sx = e->ctuX * 64 + (Z2R[sz] & 15) * 4, sy = e->ctuY * 64 + (Z2R[sz] >> 4) * 4;
      if (sx < e->cfg.width && sy < e->cfg.height) encode_cu(e, c, sz, depth + 1, lastCtuOfSlice);
    }
    return;
  }
  if (m->pred[z] == MODE_INTER) encode_cu_syntax_inter(e, c, z, depth, &e->dqpFlag); else encode_cu_syntax(e, c, z, depth);
  /* finishCU, TEncCu.cpp:1130-1147 */
  const int lastX = ((lx + size) % 64 == 0) || (lx + size == e->cfg.width), lastY = ((ty + size) % 64 == 0) || (ty + size == e->cfg.height);
  if (lastX && lastY && !lastCtuOfSlice) enc_trm(c, 0);
}

/* ============================================================================================ */
/* slice driver (TEncSlice::compressSlice, TEncSlice.cpp:640-904)                                */
/* ============================================================================================ */
void hmo_cfg_set_qp(hmo_cfg *c, int qp)
{ /* TEncSlice::initEncSlice, TEncSlice.cpp:323-352; setUpLambda :132-159 */
  c->qp = qp;
  c->lambda = 0.57 * pow(2.0, ((double)qp - 12) / 3.0);
  int qpc = CHROMA_SCALE_420[clip3(0, 57, qp)];
  c->chroma_weight = pow(2.0, (qp - qpc) / 3.0);
}

/* QpParam (TComTrQuant.cpp:71-119) of a CU at QP `qp`: what TComTrQuant::setQPforQuant hands the quantiser */
static void set_quant_qp(Enc *e, int qp)
{
  const int bdOff = 6 * (e->cfg.bit_depth - 8);
  int q = qp + bdOff; e->qpPer[0] = q / 6; e->qpRem[0] = q % 6;
  int qc = clip3(-bdOff, 57, qp);
  qc = (qc < 0) ? qc + bdOff : CHROMA_SCALE_420[qc] + bdOff;
  e->qpPer[1] = e->qpPer[2] = qc / 6; e->qpRem[1] = e->qpRem[2] = qc % 6;
}
static int compress_impl(const hmo_cfg *cfg, const uint16_t *const org[3], uint16_t *const rec[3], hmo_ctu *ctus, int maxCtus,
                         const hmo_inter_slice *hs, hmo_ctu_inter *ictus, const hmo_dqp *dq)
{
  if (!cfg || cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || (cfg->bit_depth != 8 && cfg->bit_depth != 10)) return -1;
  init_tables();
  Enc *e = (Enc *)calloc(1, sizeof(Enc));
  if (!e) return -2;
  e->cfg = *cfg;
  e->wCtu = (cfg->width + 63) / 64; e->hCtu = (cfg->height + 63) / 64;
  const int numCtus = e->wCtu * e->hCtu;
  for (int c = 0; c < 3; c++) {
    e->stride[c] = e->wCtu * (c ? 32 : 64); e->ph[c] = e->hCtu * (c ? 32 : 64);
    e->org[c] = (Pel *)calloc((size_t)e->stride[c] * e->ph[c], sizeof(Pel));
    e->rec[c] = (Pel *)calloc((size_t)e->stride[c] * e->ph[c], sizeof(Pel));
    const int w = cfg->width >> (c ? 1 : 0), h = cfg->height >> (c ? 1 : 0);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) e->org[c][y * e->stride[c] + x] = (Pel)org[c][y * w + x];
    e->coef[c] = (TCoeff *)calloc((size_t)numCtus * (c ? 1024 : 4096), sizeof(TCoeff));
  }
  e->meta = (CtuMeta *)calloc(numCtus, sizeof(CtuMeta));
  InterSlice islice; RefPic refPics[32]; int numRefPics = 0;
  g_slice_type = I_SLICE;
  if (hs) { /* slice header / DPB view of a P slice */
    if (hs->slice_type != P_SLICE && hs->slice_type != B_SLICE) { free(e); return -3; }
    memset(&islice, 0, sizeof(islice));
    islice.sliceType = hs->slice_type; islice.poc = hs->poc;
    islice.colFromL0 = hs->col_from_l0; islice.colRefIdx = hs->col_ref_idx; islice.tmvp = hs->tmvp; islice.mvdL1Zero = hs->mvd_l1_zero;
    islice.maxMergeCand = hs->max_merge_cand; islice.checkLDC = hs->check_ldc;
    islice.lambdaMotionSAD = hs->lambda_motion_sad; islice.lambdaMotionSSE = hs->lambda_motion_sse;
    const hmo_ref_pic *seen[32];
    for (int l = 0; l < 2; l++) {
      islice.numRefIdx[l] = hs->num_ref_idx[l];
      for (int i = 0; i < hs->num_ref_idx[l]; i++) {
        const hmo_ref_pic *hp = hs->ref[l][i]; int k;
        for (k = 0; k < numRefPics; k++) if (seen[k] == hp) break;
        if (k == numRefPics) {
          RefPic *rp = &refPics[numRefPics]; seen[numRefPics++] = hp;
          memset(rp, 0, sizeof(*rp));
          rp->poc = hp->poc; rp->isLongTerm = hp->long_term; rp->sliceType = hp->slice_type;
          refpic_extend(rp, hp->plane, cfg->width, cfg->height);
          rp->predMode = hp->pred_mode;
          for (int ll = 0; ll < 2; ll++) { rp->mv[ll] = hp->mv[ll]; rp->refIdx[ll] = hp->ref_idx[ll]; rp->numRef[ll] = hp->num_ref[ll]; memcpy(rp->refPoc[ll], hp->ref_poc[ll], sizeof(rp->refPoc[ll])); memcpy(rp->refLT[ll], hp->ref_lt[ll], sizeof(rp->refLT[ll])); }
        }
        islice.ref[l][i] = &refPics[k];
      }
    }
    for (int i1 = 0; i1 < islice.numRefIdx[1]; i1++) {
      islice.list1ToList0[i1] = -1;
      for (int i0 = 0; i0 < islice.numRefIdx[0]; i0++) if (islice.ref[0][i0]->poc == islice.ref[1][i1]->poc) { islice.list1ToList0[i1] = i0; break; }
    }
    e->is = &islice;
    g_slice_type = hs->cabac_init_type;
  }
  e->lambda = cfg->lambda; e->sqrtLambda = sqrt(cfg->lambda);
  e->chromaWeight = cfg->chroma_weight; e->lambdaC = cfg->lambda / cfg->chroma_weight;
  set_quant_qp(e, cfg->qp);
  e->ctuQp = e->refQp = cfg->qp;
  e->dq = (dq && dq->use_dqp) ? dq : NULL;
  e->dqpFlag = e->dq ? (dq->dqp_flag_in != 0) : 0;
  e->lastQp = (int8_t *)calloc(numCtus, 1);
  cabac_init(&e->slot[0][CI_CURR_BEST], cfg->qp);
  const int limit = (maxCtus > 0 && maxCtus < numCtus) ? maxCtus : numCtus;
  for (int a = 0; a < limit; a++) {
    e->ctuAddr = a; e->ctuX = a % e->wCtu; e->ctuY = a / e->wCtu;
    e->cm = e->meta + a; for (int c = 0; c < 3; c++) e->cc[c] = e->coef[c] + (size_t)a * (c ? 1024 : 4096);
    { /* TComDataCU::initCtu, TComDataCU.cpp:357-470 */
      CtuMeta *m = e->cm;
      memset(m->depth, 0, 256); memset(m->part, SIZE_NONE, 256); memset(m->pred, MODE_NONE, 256);
      memset(m->dirL, DC_IDX, 256); memset(m->dirC, 0, 256); memset(m->tr, 0, 256);
      for (int c = 0; c < 3; c++) { memset(m->cbf[c], 0, 256); memset(m->ts[c], 0, 256); }
      memset(m->skip, 0, 256); memset(m->mrg, 0, 256); memset(m->mrgIdx, 0, 256); memset(m->interDir, 0, 256);
      for (int l = 0; l < 2; l++) { memset(m->mv[l], 0, sizeof(m->mv[l])); memset(m->mvd[l], 0, sizeof(m->mvd[l])); memset(m->refIdx[l], -1, 256); memset(m->mvpIdx[l], -1, 256); memset(m->mvpNum[l], -1, 256); }
      memset(m->qp, cfg->qp, 256);
    }
    if (e->dq) {
      /* the QP of this CTU (TEncCu::xComputeQP :1154 / the rate control's, TEncSlice.cpp:767-808) and its predictor: both neighbouring quantisation
         groups lie outside the CTU, so TComDataCU::getRefQP (:1413) is the QP of the last CU coded before it in this CTU row / slice (:1434-1468) */
      e->ctuQp = dq->ctu_qp ? dq->ctu_qp[a] : cfg->qp;
      e->refQp = (a == 0 || (e->ctuX == 0 && cfg->wpp)) ? cfg->qp : e->lastQp[a - 1];
      set_quant_qp(e, e->ctuQp);
      if (dq->ctu_lambda) { /* TComRdCost::setLambda (TComRdCost.cpp:194-216) + TComTrQuant::setLambdas with the chroma weight of the slice, TEncSlice.cpp:793-803 */
        const double L = dq->ctu_lambda[a];
        e->lambda = L; e->sqrtLambda = sqrt(L); e->lambdaC = L / e->chromaWeight;
        if (e->is) { islice.lambdaMotionSAD = (uint32_t)floor(65536.0 * e->sqrtLambda); islice.lambdaMotionSSE = (uint32_t)floor(65536.0 * L); }
      }
    }
    if (a == 0) cabac_init(&e->slot[0][CI_CURR_BEST], cfg->qp);
    else if (e->ctuX == 0 && cfg->wpp) {
      cabac_init(&e->slot[0][CI_CURR_BEST], cfg->qp);
      if (e->ctuY > 0 && e->wCtu > 1) { uint64_t f = e->slot[0][CI_CURR_BEST].frac; e->slot[0][CI_CURR_BEST] = e->wppSync; e->slot[0][CI_CURR_BEST].frac = f; }
    }
    e->cur = e->slot[0][CI_CURR_BEST];
    double cost; uint32_t bits, dist;
    compress_cu(e, 0, 0, SIZE_NONE, &cost, &bits, &dist);
    hmo_ctu *o = ctus + a;
    o->total_cost = cost; o->total_bits = bits; o->total_dist = dist;
    /* TEncCu::encodeCtu on m_pppcRDSbacCoder[0][CI_CURR_BEST], TEncSlice.cpp:818-825 */
    reset_bits(&e->slot[0][CI_CURR_BEST]);
    if (e->dq) e->dqpFlag = 1;                                   /* TEncCu::encodeCtu :358-361 */
    encode_cu(e, &e->slot[0][CI_CURR_BEST], 0, 0, a == numCtus - 1);
    if (e->ctuX == 1 && cfg->wpp) e->wppSync = e->slot[0][CI_CURR_BEST];
    { int z = 255; while (z > 0 && e->cm->pred[z] == MODE_NONE) z--; e->lastQp[a] = e->cm->qp[z]; }   /* getLastValidPartIdx :1421 */
    if (dq && dq->qp_out) memcpy(dq->qp_out + (size_t)a * 256, e->cm->qp, 256);
  }
  if (dq && dq->dqp_flag_out) *dq->dqp_flag_out = e->dqpFlag;
  free(e->lastQp);
  for (int a = 0; a < limit; a++) {
    const CtuMeta *m = e->meta + a; hmo_ctu *o = ctus + a;
    memcpy(o->depth, m->depth, 256); memcpy(o->part_size, m->part, 256); memcpy(o->pred_mode, m->pred, 256);
    memcpy(o->intra_dir_luma, m->dirL, 256); memcpy(o->intra_dir_chroma, m->dirC, 256); memcpy(o->tr_idx, m->tr, 256);
    for (int c = 0; c < 3; c++) { memcpy(o->cbf[c], m->cbf[c], 256); memcpy(o->tskip[c], m->ts[c], 256); }
    memcpy(o->coeff_y, e->coef[0] + (size_t)a * 4096, sizeof(TCoeff) * 4096);
    memcpy(o->coeff_cb, e->coef[1] + (size_t)a * 1024, sizeof(TCoeff) * 1024);
    memcpy(o->coeff_cr, e->coef[2] + (size_t)a * 1024, sizeof(TCoeff) * 1024);
    if (ictus) {
      hmo_ctu_inter *io = ictus + a;
      memcpy(io->skip, m->skip, 256); memcpy(io->merge_flag, m->mrg, 256); memcpy(io->merge_idx, m->mrgIdx, 256); memcpy(io->inter_dir, m->interDir, 256);
      for (int l = 0; l < 2; l++) {
        for (int z = 0; z < 256; z++) { io->mv[l][z][0] = m->mv[l][z].x; io->mv[l][z][1] = m->mv[l][z].y; io->mvd[l][z][0] = m->mvd[l][z].x; io->mvd[l][z][1] = m->mvd[l][z].y; }
        memcpy(io->ref_idx[l], m->refIdx[l], 256); memcpy(io->mvp_idx[l], m->mvpIdx[l], 256); memcpy(io->mvp_num[l], m->mvpNum[l], 256);
      }
    }
  }
  for (int c = 0; c < 3; c++) {
    const int w = cfg->width >> (c ? 1 : 0), h = cfg->height >> (c ? 1 : 0);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) rec[c][y * w + x] = (uint16_t)e->rec[c][y * e->stride[c] + x];
    free(e->org[c]); free(e->rec[c]); free(e->coef[c]);
  }
  for (int k = 0; k < numRefPics; k++) for (int c = 0; c < 3; c++) free(refPics[k].buf[c]);
  g_slice_type = I_SLICE;
  free(e->meta); free(e);
  return 0;
}

#include "hm_oracle_dbk.inc"
#include "hm_oracle_sao.inc"
#include "hm_oracle_bits.inc"
#include "hm_oracle_yuv.inc"

int hmo_compress_slice(const hmo_cfg *cfg, const uint16_t *const org[3], uint16_t *const rec[3], hmo_ctu *ctus)
{ return compress_impl(cfg, org, rec, ctus, 0, NULL, NULL, NULL); }
int hmo_compress_rows(const hmo_cfg *cfg, const uint16_t *const org[3], uint16_t *const rec[3], hmo_ctu *ctus, int max_ctus)
{ return compress_impl(cfg, org, rec, ctus, max_ctus, NULL, NULL, NULL); }
int hmo_compress_slice_inter(const hmo_cfg *cfg, const hmo_inter_slice *slice, const uint16_t *const org[3], uint16_t *const rec[3],
                             hmo_ctu *ctus, hmo_ctu_inter *ictus)
{ return slice ? compress_impl(cfg, org, rec, ctus, 0, slice, ictus, NULL) : -1; }
int hmo_compress_slice_dqp(const hmo_cfg *cfg, const hmo_inter_slice *slice, const uint16_t *const org[3], uint16_t *const rec[3],
                           hmo_ctu *ctus, hmo_ctu_inter *ictus, const hmo_dqp *dq)
{ return compress_impl(cfg, org, rec, ctus, 0, slice, ictus, dq); }

/* TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64-139), layer 0 (one unit per CTU): activity = 1 + the smallest "variance" of the unit's four
   quadrants, where -- as in the reference -- every quadrant's sums are divided by the sample count of the WHOLE unit; the average is the
   running sum over the units in raster order divided by their number */
void hmo_preanalyze(const uint16_t *luma, int width, int height, double *activity, double *avg_activity)
{
  const int wU = (width + 63) / 64, hU = (height + 63) / 64;
  double sumAct = 0.0;
  for (int uy = 0; uy < hU; uy++) for (int ux = 0; ux < wU; ux++) {
    const int x0 = ux * 64, y0 = uy * 64, w = width - x0 < 64 ? width - x0 : 64, h = height - y0 < 64 ? height - y0 : 64;
    uint64_t sum[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0}; unsigned n = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++, n++) {
      const int k = (y >= (h >> 1) ? 2 : 0) + (x >= (w >> 1) ? 1 : 0);
      const int v = (int16_t)luma[(size_t)(y0 + y) * width + x0 + x];
      sum[k] += (uint64_t)(int64_t)v; sq[k] += (uint64_t)(int64_t)(v * v);
    }
    double minVar = 1.7976931348623157e308;
    for (int k = 0; k < 4; k++) {
      const double average = (double)sum[k] / n;
      const double variance = (double)sq[k] / n - average * average;
      if (variance < minVar) minVar = variance;
    }
    activity[uy * wU + ux] = 1.0 + minVar;
    sumAct += activity[uy * wU + ux];
  }
  *avg_activity = sumAct / (wU * hU);
}
/* TEncCu::xComputeQP, TEncCu.cpp:1154-1176 */
int hmo_aq_qp(double activity, double avg_activity, int aq_range, int slice_qp, int bit_depth)
{
  const double maxQScale = pow(2.0, aq_range / 6.0);
  const double normAct = (maxQScale * activity + avg_activity) / (activity + maxQScale * avg_activity);
  const double qpOffset = log(normAct) / log(2.0) * 6.0;
  return clip3(-6 * (bit_depth - 8), 51, slice_qp + (int)floor(qpOffset + 0.49999));
}
