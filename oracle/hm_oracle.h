/* TEST INFRASTRUCTURE ONLY -- CPU restatement (the "oracle") of the hot path of liron88/HM-16.2:
 * TEncSlice::compressSlice and everything below it for I slices.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 * The product library (hm-16.2_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * Parity status: PINNED. The restatement is checked bit-for-bit against the real reference
 * (oracle/_ref, built from /root/reference by oracle/Makefile.ref) through the fixtures in
 * tests/golden/ that tests/gen_golden.py produced with oracle/_ref/hm_dump.
 */
#ifndef HM_ORACLE_H
#define HM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sequence/slice parameters the reference's compressSlice reads through its members
 * (SURVEY.md 8b "Inputs"): all-intra configuration of cfg/encoder_intra_main*.cfg. */
typedef struct {
  int width, height;        /* luma samples, multiples of 8 */
  int bit_depth;            /* 8 or 10, luma == chroma */
  int qp;                   /* slice QP (TEncSlice::initEncSlice, TEncSlice.cpp:293) */
  int wpp;                  /* WaveFrontSynchro: CABAC sync per CTU row (TEncSlice.cpp:740-755,855-858) */
  double lambda;            /* TComRdCost::m_dLambda (TEncSlice.cpp:323-352 -> setUpLambda :132) */
  double chroma_weight;     /* TComRdCost::m_distortionWeight[Cb/Cr] (TEncSlice.cpp:146) */
} hmo_cfg;

/* Fills lambda and chroma_weight from qp exactly as TEncSlice::initEncSlice does for an I slice of an
 * all-intra GOP (GOPSize 1, depth 0): lambda = 0.57 * 2^((qp-12)/3). */
void hmo_cfg_set_qp(hmo_cfg *c, int qp);

/* Per-CTU output, same fields and order as TComDataCU's per-partition arrays (256 4x4 partitions in
 * z-scan order, TComDataCU.h:86-157) and the coefficient packing of m_pcTrCoeff (TU at zorder*16). */
typedef struct {
  double total_cost;
  uint32_t total_bits, total_dist;
  uint8_t depth[256], part_size[256], pred_mode[256], intra_dir_luma[256], intra_dir_chroma[256],
          tr_idx[256], cbf[3][256], tskip[3][256];
  int32_t coeff_y[4096], coeff_cb[1024], coeff_cr[1024];
} hmo_ctu;

/* The restated TEncSlice::compressSlice for one I picture.
 *   org[3]  : original planes, uint16 samples, tightly packed (w x h, w/2 x h/2 x 2)
 *   rec[3]  : reconstructed (pre-deblocking) planes out, same layout
 *   ctus    : ceil(w/64)*ceil(h/64) records, raster order
 * Returns 0, or a negative value on bad arguments. */
int hmo_compress_slice(const hmo_cfg *cfg, const uint16_t *const org[3], uint16_t *const rec[3], hmo_ctu *ctus);

/* Same, restricted to CTU rows [row0, row1) -- usable only when wpp==0 is false?  No: rows depend on
 * each other, so this variant needs rec/ctus of rows < row0 already filled (used for bounded timing). */
int hmo_compress_rows(const hmo_cfg *cfg, const uint16_t *const org[3], uint16_t *const rec[3], hmo_ctu *ctus,
                      int max_ctus);

/* ---- P and B slices (rows a2 inter, a4-a7, a12, a14 of SURVEY.md section 8): reference pictures and slice parameters are inputs,
 * exactly what TEncSlice::compressSlice finds in the slice header / decoded picture buffer ---- */
typedef struct {
  int poc, slice_type, long_term;
  const uint16_t *plane[3];          /* final reconstruction (after the loop filters), tightly packed */
  const uint8_t *pred_mode;          /* motion field after TComPic::compressMotion: [numCtus*256] */
  const int16_t *mv[2];              /* [numCtus*256*2] (x, y) */
  const int8_t *ref_idx[2];          /* [numCtus*256] */
  int num_ref[2], ref_poc[2][16], ref_lt[2][16];    /* the reference lists that picture was coded with */
} hmo_ref_pic;
typedef struct {
  int slice_type, poc;               /* 1 = P, 0 = B */
  int cabac_init_type;               /* context table actually used (TEncSbac::resetEntropy :106-115): 0 = B, 1 = P */
  int num_ref_idx[2];
  const hmo_ref_pic *ref[2][16];
  int col_from_l0, col_ref_idx, tmvp, mvd_l1_zero, max_merge_cand, check_ldc;
  uint32_t lambda_motion_sad, lambda_motion_sse;    /* TComRdCost::m_uiLambdaMotionSAD/SSE[0] */
} hmo_inter_slice;
typedef struct {
  uint8_t skip[256], merge_flag[256], merge_idx[256], inter_dir[256];
  int16_t mv[2][256][2], mvd[2][256][2];
  int8_t ref_idx[2][256], mvp_idx[2][256], mvp_num[2][256];
} hmo_ctu_inter;
int hmo_compress_slice_inter(const hmo_cfg *cfg, const hmo_inter_slice *slice, const uint16_t *const org[3], uint16_t *const rec[3],
                             hmo_ctu *ctus, hmo_ctu_inter *ictus);

/* ---- cu_qp_delta: adaptive QP and rate control hand compressSlice a QP per CTU (MaxCuDQPDepth 0: the CTU is the
 * quantisation group; MaxDeltaQP 0).  The search then quantises every CU of the CTU at that QP, prices the delta QP where the reference
 * does (TEncCu::xCheckDQP :1742, the split candidate :1052-1085, and -- while TEncCu::m_bEncodeDQP happens to be set -- with the first
 * coded block of an intra candidate, TEncCu.cpp:1629-1633) and leaves TComDataCU::m_phQP behind. ---- */
typedef struct {
  int use_dqp;                       /* PPS cu_qp_delta_enabled_flag */
  int dqp_flag_in;                   /* TEncCu::m_bEncodeDQP on entry: what the previous picture's encodeSlice left */
  const int8_t *ctu_qp;              /* [numCtus] QP of each CTU (TEncCu::xComputeQP / TEncRateCtrl::getRCQP); NULL: the slice QP */
  int8_t *qp_out;                    /* [numCtus*256] m_phQP as compressSlice leaves it; may be NULL */
  int *dqp_flag_out;                 /* m_bEncodeDQP on exit; may be NULL */
  const double *ctu_lambda;          /* [numCtus] lambda of each CTU's search: the LCU-level rate control sets one per CTU (TEncSlice.cpp:776-808: m_pcRdCost->setLambda,
                                        m_pcTrQuant->setLambdas); NULL: the slice lambda.  n4 stage 2, oracle only so far */
} hmo_dqp;
/* slice NULL: I slice */
int hmo_compress_slice_dqp(const hmo_cfg *cfg, const hmo_inter_slice *slice, const uint16_t *const org[3], uint16_t *const rec[3],
                           hmo_ctu *ctus, hmo_ctu_inter *ictus, const hmo_dqp *dq);
/* TEncPreanalyzer::xPreanalyze for layer 0 (one unit per CTU): activity[numCtus] and their average; hmo_aq_qp = TEncCu::xComputeQP */
void hmo_preanalyze(const uint16_t *luma, int width, int height, double *activity, double *avg_activity);
int hmo_aq_qp(double activity, double avg_activity, int aq_range, int slice_qp, int bit_depth);

/* ---- deblocking filter (TComLoopFilter::loopFilterPic, TComLoopFilter.cpp:130-158) on the picture compressSlice left:
 * rec is filtered in place (vertical edges of the whole picture, then horizontal edges).  cfg->qp = slice QP; ref_poc = POCs of the
 * slice's reference pictures [list][idx] (boundary strength compares pictures); ictus may be NULL for an I slice. ---- */
int hmo_deblock(const hmo_cfg *cfg, int slice_type, const int32_t ref_poc[2][16], const hmo_ctu *ctus, const hmo_ctu_inter *ictus,
                uint16_t *const rec[3]);

/* ---- sample adaptive offset, encoder side (TEncSampleAdaptiveOffset::SAOProcess, TEncGOP.cpp:1483) on the deblocked picture:
 * rec is replaced by the SAO output.  cfg: qp, lambda, chroma_weight of the slice; depth = temporal depth of the picture;
 * disabled_rate[comp][depth] = m_saoDisabledRate, read for depth-1 and written for depth (carried from picture to picture by the caller);
 * sao_params (may be NULL) receives numCtus x 3 x 35 int32: modeIdc, typeIdc, typeAuxInfo, offset[32] as coded. ---- */
int hmo_sao(const hmo_cfg *cfg, int cabac_init_type, int depth, double disabled_rate[3][8], const uint16_t *const org[3], uint16_t *const rec[3],
            int32_t *sao_params, int32_t enabled_out[3]);

/* ---- bitstream pass (TEncSlice::encodeSlice, TEncSlice.cpp:910-1095): the CABAC-coded slice data of one picture from what
 * compressSlice (ctus / ictus) and SAO (sao: numCtus x 3 x 35 int32 as hmo_sao writes them, or NULL) left.  cfg: width, height, bit_depth, wpp.
 * out receives the substreams back to back (one per CTU row with wpp, else one), sub_sizes their byte counts; each substream ends with the
 * terminating bin, the CABAC flush and the byte alignment, exactly the bytes TEncGOP concatenates behind the slice header (before emulation
 * prevention).  next_cabac_init_type = what determineCabacInitIdx (TEncSbac.cpp:163) leaves in the PPS for the following pictures. ---- */
typedef struct {
  int slice_type;                    /* 2 = I, 1 = P, 0 = B */
  int qp;
  int cabac_init_type;               /* context table of a P / B slice (TEncSbac::resetEntropy :106-115) */
  int num_ref_idx[2], mvd_l1_zero, max_merge_cand;
  int sao_enabled[2];                /* slice_sao_luma_flag, slice_sao_chroma_flag */
} hmo_bits_slice;
int hmo_encode_slice(const hmo_cfg *cfg, const hmo_bits_slice *slice, const hmo_ctu *ctus, const hmo_ctu_inter *ictus, const int32_t *sao,
                     uint8_t *out, size_t out_cap, uint32_t *sub_sizes, int *next_cabac_init_type, uint32_t *num_bins);

/* ---- picture ingest and output (TVideoIOYuv::read / ::write, TVideoIOYuv.cpp:633-792) for planar 4:2:0 files.
 * read: one frame of a file_w x file_h file (8-bit samples, or 16-bit little endian when file_bit_depth > 8) into tightly packed planes of
 * (file_w + pad_x) x (file_h + pad_y), padded by repetition and scaled to internal_bit_depth.  write: the planes minus the conformance window
 * at the right / bottom, scaled to file_bit_depth (rounded and clipped when that is lower). ---- */
int hmo_yuv_read(const uint8_t *file, int file_w, int file_h, int file_bit_depth, int internal_bit_depth, int pad_x, int pad_y, uint16_t *const planes[3]);
int hmo_yuv_write(const uint16_t *const planes[3], int width, int height, int internal_bit_depth, int file_bit_depth, int crop_right, int crop_bottom, uint8_t *file);

/* ---- primitives, exported for the known-answer tests (TComRdCost.cpp / TComTrQuant.cpp) ---- */
uint32_t hmo_sad(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int sub_shift, int bit_depth);
uint32_t hmo_sse(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int bit_depth);
uint32_t hmo_hads(const int16_t *org, int so, const int16_t *cur, int sc, int w, int h, int bit_depth);
void hmo_fwd_transform(int bit_depth, const int32_t *block, int32_t *coeff, int n, int use_dst);
void hmo_inv_transform(int bit_depth, const int32_t *coeff, int32_t *block, int n, int use_dst);

/* optional trace of every RD cost evaluation ("RD bits dist cost"), same text as oracle/_ref/hm_dump's HM_TRACE */
void hmo_set_trace(const char *path);

#ifdef __cplusplus
}
#endif
#endif
