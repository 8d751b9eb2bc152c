/* hm355 -- MI355X-native replacement for the CTU-level RD search of HM-16.2 (liron88/HM-16.2).
 *
 * C ABI of the drop-in boundary.  The reference has no FFI layer; the seam is the C++ member
 *     Void TEncSlice::compressSlice( TComPic* pcPic )      source/Lib/TLibEncoder/TEncSlice.h:118
 *                                                           (body TEncSlice.cpp:640-904, caller TEncGOP.cpp:1138)
 * The entry points below are what a binding of that member would call: plain pointers and sizes,
 * caller-owned buffers, blocking calls, negative return codes instead of assert/exit.
 * INTEGRATION.md shows the few lines a reference maintainer adds inside TEncSlice::compressSlice.
 *
 * All work is done by hand-written HIP kernels (hm-16.2_amd/csrc/hm355.hip + hm355_core.h).  There is NO
 * CPU fallback: without a usable gfx950 device hm355_create() fails with HM355_ERR_NO_DEVICE.
 */
#ifndef HM355_H
#define HM355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM355_OK               0
#define HM355_ERR_ARG         -1   /* bad argument / unsupported configuration */
#define HM355_ERR_NO_DEVICE   -2   /* no HIP device or kernel image not loadable */
#define HM355_ERR_NOMEM       -3
#define HM355_ERR_DEVICE      -4   /* a HIP call or a kernel failed */

/* Sequence-level parameters: the SPS/PPS/TEncCfg getters compressSlice reads (TEncCfg.h).  The block structure is the one every cfg
 * file of the reference uses (cfg/encoder_intra_main*.cfg, encoder_lowdelay_P_main.cfg, encoder_lowdelay_main.cfg,
 * encoder_randomaccess_main*.cfg: CTU 64, depth 4, TU 4..32, RQT depth 3); other values are rejected with HM355_ERR_ARG rather than
 * silently ignored.  I, P and B slices are all accepted (the slice type is a per-call parameter). */
typedef struct {
  int32_t width, height;          /* SourceWidth/Height, multiples of 8 (min CU) */
  int32_t bit_depth;              /* InternalBitDepth 8 or 10 (luma == chroma) */
  int32_t ctu_size;               /* MaxCUWidth = 64 */
  int32_t max_cu_depth;           /* MaxPartitionDepth = 4 */
  int32_t tu_log2_max, tu_log2_min;   /* 5, 2 */
  int32_t tu_max_depth_intra;     /* QuadtreeTUMaxDepthIntra = 3 */
  int32_t wavefront_synchro;      /* WaveFrontSynchro 0/1 (TEncSlice.cpp:740-755,855-858) */
  int32_t max_batch;              /* how many independent pictures one call may carry (>=1) */
} hm355_seq_cfg;

/* Slice-level parameters: what TEncSlice::initEncSlice / setUpLambda (TEncSlice.cpp:132-159,180-481)
 * push into TComRdCost / TComTrQuant before compressSlice runs. */
typedef struct {
  int32_t slice_type;             /* 2 = I_SLICE in the I-slice entry points; the inter entry points (hm355_inter_slice_desc::base) take 1 = P_SLICE or 0 = B_SLICE */
  int32_t qp;                     /* TComSlice::getSliceQp() */
  double  lambda;                 /* TComRdCost::m_dLambda */
  double  chroma_weight;          /* TComRdCost::m_distortionWeight[Cb]==[Cr] */
} hm355_slice_desc;

/* planar 4:2:0 picture, 16-bit samples, tightly packed rows (stride == width of the plane) */
typedef struct {
  uint16_t *plane[3];
} hm355_planes;

/* Per-CTU result: the TComDataCU arrays compressSlice leaves in pcPic->getCtu(rs)
 * (TComDataCU.h:86-157), 256 4x4 partitions in z-scan order, and m_pcTrCoeff in HM's TU packing
 * (TU with z-order index z starts at z*16 for luma, z*4 for chroma; row-major inside the TU). */
typedef struct {
  double   total_cost;            /* m_dTotalCost */
  uint32_t total_bits;            /* m_uiTotalBits */
  uint32_t total_dist;            /* m_uiTotalDistortion */
  uint8_t  depth[256], part_size[256], pred_mode[256], intra_dir_luma[256], intra_dir_chroma[256],
           tr_idx[256], cbf[3][256], tskip[3][256];
  int32_t  coeff_y[4096], coeff_cb[1024], coeff_cr[1024];
} hm355_ctu_out;

/* m_uiPicTotalBits / m_dPicRdCost / m_uiPicDist (TEncSlice.cpp:889-891) */
typedef struct {
  uint64_t pic_total_bits;
  double   pic_rd_cost;
  uint64_t pic_dist;
} hm355_slice_stats;

typedef struct hm355_ctx hm355_ctx;

/* TEncTop::create/init equivalent for the hot path: allocates every device buffer once. */
int  hm355_create(const hm355_seq_cfg *cfg, hm355_ctx **out);
void hm355_destroy(hm355_ctx *ctx);
const char *hm355_last_error(const hm355_ctx *ctx);
/* identifies the kernel source this library was built from (hash of hm-16.2_amd/csrc + this header): measurement records carry it */
const char *hm355_build_id(void);

/* Replacement of TEncSlice::compressSlice for one picture (host buffers in, host buffers out).
 *   org  : TComPic::getPicYuvOrg()          rec : TComPic::getPicYuvRec() (pre-deblocking)
 *   ctus : one record per CTU, raster order  stats : may be NULL */
int hm355_compress_slice(hm355_ctx *ctx, const hm355_slice_desc *slice, const hm355_planes *org,
                         hm355_planes *rec, hm355_ctu_out *ctus, hm355_slice_stats *stats);

/* The same for n independent pictures (all-intra: every picture is an IDR-like I slice with its own
 * CABAC reset, TEncSlice.cpp:653-654), evaluated concurrently.  n <= max_batch. */
int hm355_compress_slices(hm355_ctx *ctx, int n, const hm355_slice_desc *slices, const hm355_planes *org,
                          hm355_planes *rec, hm355_ctu_out *const *ctus, hm355_slice_stats *stats);

/* ---- P and B slices (encoder_lowdelay_P_main.cfg, encoder_lowdelay_main.cfg, encoder_randomaccess_main*.cfg): the reference
 * pictures and slice-header values that TEncSlice::compressSlice finds through pcSlice->getRefPic(list, idx) / TComSlice
 * getters are inputs.  Replaces the same member for inter slices: merge / skip, AMVP, TZ integer search + fractional
 * refinement (TEncSearch::predInterSearch / xMotionEstimation, TEncSearch.cpp:3075-3906), bi-prediction search
 * (xPatternSearch :3932, GPB shortcut, mvd_l1_zero), AMP, inter RQT (xEstimateResidualQT :4680) and the intra candidates of
 * inter slices.  Weighted prediction off, FEN on (one bi-prediction iteration), as in the reference's cfg files. ---- */
typedef struct {
  int32_t poc, slice_type, long_term;
  const uint16_t *plane[3];           /* TComPic::getPicYuvRec() after the loop filters, tightly packed */
  const uint8_t *pred_mode;           /* motion field after TComPic::compressMotion: [numCtus*256] */
  const int16_t *mv[2];               /* [numCtus*256*2] (x, y), quarter samples */
  const int8_t *ref_idx[2];           /* [numCtus*256] */
  int32_t num_ref[2], ref_poc[2][16], ref_lt[2][16];   /* reference lists that picture was coded with (TMVP scaling) */
} hm355_ref_pic;
struct hm355_ref;
typedef struct {
  hm355_slice_desc base;              /* slice_type 1 (P) or 0 (B), qp, lambda, chroma weight */
  int32_t poc;
  int32_t cabac_init_type;            /* context table in use (TEncSbac::resetEntropy, TEncSbac.cpp:106-115): 0 = B, 1 = P */
  int32_t num_ref_idx[2];
  const hm355_ref_pic *ref[2][16];
  int32_t col_from_l0, col_ref_idx, tmvp, mvd_l1_zero, max_merge_cand, check_ldc;
  uint32_t lambda_motion_sad, lambda_motion_sse;       /* TComRdCost::m_uiLambdaMotionSAD / SSE[0] */
  const struct hm355_ref *dev_ref[2][16];              /* device-resident alternative to ref[l][i] (hm355_ref_from_slot); used when non-NULL */
} hm355_inter_slice_desc;
/* per-CTU motion data: m_skipFlag, m_pbMergeFlag, m_puhMergeIndex, m_puhInterDir, m_acCUMvField[2] (mv, mvd, refIdx),
 * m_apiMVPIdx / m_apiMVPNum (TComDataCU.h:86-157) */
typedef struct {
  uint8_t skip[256], merge_flag[256], merge_idx[256], inter_dir[256];
  int16_t mv[2][256][2], mvd[2][256][2];
  int8_t  ref_idx[2][256], mvp_idx[2][256], mvp_num[2][256];
} hm355_ctu_inter_out;
int hm355_compress_slice_inter(hm355_ctx *ctx, const hm355_inter_slice_desc *slice, const hm355_planes *org,
                               hm355_planes *rec, hm355_ctu_out *ctus, hm355_ctu_inter_out *ictus, hm355_slice_stats *stats);
/* n P pictures that do not reference each other (the current pictures of n streams, or of n independent GOP chains),
 * evaluated concurrently; a hm355_ref_pic named by several slices is uploaded once.  n <= max_batch. */
int hm355_compress_slices_inter(hm355_ctx *ctx, int n, const hm355_inter_slice_desc *slices, const hm355_planes *org,
                                hm355_planes *rec, hm355_ctu_out *const *ctus, hm355_ctu_inter_out *const *ictus, hm355_slice_stats *stats);

/* ---- device-resident reference pictures: the finished picture of a slot (after hm355_deblock_run) becomes a reference without a
 * host round trip: border extension (TComPicYuv::extendPicBorder, TComPicYuv.cpp:171) and TComPic::compressMotion (TEncGOP.cpp:1660)
 * run on the device.  is_inter = 0 for an I picture (no motion), else the slot's motion data of its last hm355_compress_slices_inter;
 * num_ref / ref_poc / ref_lt = the reference lists that picture was coded with (TMVP scaling).  The slot can be reused afterwards. ---- */
typedef struct hm355_ref hm355_ref;
int hm355_ref_from_slot(hm355_ctx *ctx, int slot, int32_t poc, int32_t is_inter, const int32_t num_ref[2], const int32_t ref_poc[2][16],
                        const int32_t ref_lt[2][16], hm355_ref **out);
void hm355_ref_release(hm355_ctx *ctx, hm355_ref *ref);
/* The same reference picture as ONE blob of hm355_ref_bytes() bytes (header, border-extended planes, compressed motion field), in host or
 * device memory: what crosses devices when pictures of one temporal layer are searched on different GPUs and the finished pictures are
 * all-gathered (SURVEY.md 8e "Inter"; TEncGOP.cpp:1184,1483,1660 make a picture a reference only after deblocking, SAO and
 * compressMotion, which is the state hm355_ref_from_slot captures).  A device buffer goes to RCCL as it is (hm-16.2_amd/gop_shard.py).
 * user[4]: four doubles that travel with the picture (the scheduler's per-picture state, e.g. the SAO disabled rates); may be NULL. */
size_t hm355_ref_bytes(const hm355_ctx *ctx);
int hm355_ref_export(hm355_ctx *ctx, const hm355_ref *ref, void *buf, const double user[4]);
int hm355_ref_import(hm355_ctx *ctx, const void *buf, hm355_ref **out, double user[4]);

/* ---- deblocking filter: TComLoopFilter::loopFilterPic (TLibCommon/TComLoopFilter.cpp:130-158), the step TEncGOP runs after
 * compressSlice (TEncGOP.cpp:1184) to turn the reconstruction into a reference picture.  Deblocking offsets 0, one slice,
 * no PCM / lossless (every cfg of the reference); SAO is a separate, later stage. ---- */
typedef struct {
  int32_t slice_type, qp;             /* slice QP (no delta QP: every CU carries it) */
  int32_t ref_poc[2][16];             /* POC of the slice's reference pictures [list][idx]: boundary strength compares pictures */
} hm355_dbk_desc;
/* host buffers: rec (the pre-deblocking reconstruction compressSlice left) is filtered in place, using ctus / ictus of the same slice
 * (ictus may be NULL for an I slice) */
int hm355_deblock(hm355_ctx *ctx, const hm355_dbk_desc *desc, const hm355_ctu_out *ctus, const hm355_ctu_inter_out *ictus, hm355_planes *rec);
/* device-resident: filters the reconstruction of slots 0..n-1 in place, right after hm355_run / hm355_compress_slices(_inter) left
 * their per-CTU data there; hm355_download then returns the deblocked picture */
int hm355_deblock_run(hm355_ctx *ctx, int n, const hm355_dbk_desc *descs);

/* ---- sample adaptive offset, encoder side: TEncSampleAdaptiveOffset::SAOProcess (TEncGOP.cpp:1483) on the deblocked pictures of slots
 * 0..n-1 (after hm355_deblock_run), against the originals uploaded to the same slots: statistics, picture-level on/off, per-CTU
 * off / new / merge decision with the SAO syntax on the CABAC estimator, offsets applied in place.  One slice, no tiles,
 * SAOLcuBoundary 0, offset bit shifts 0 (every cfg of the reference). ---- */
typedef struct {
  int32_t qp, cabac_init_type, depth;  /* slice QP; context table of the slice (0 B, 1 P, 2 I); temporal depth (TComSlice::getDepth) */
  double lambda, chroma_weight;        /* TComSlice::getLambdas(): luma lambda, chroma lambda = lambda / chroma_weight */
  double disabled_rate[3][8];          /* in/out: TEncSampleAdaptiveOffset::m_saoDisabledRate[component][depth], carried from picture to picture */
  int32_t enabled[3];                  /* out: slice-level SAO flags (luma, Cb, Cr) */
  int32_t *params;                     /* out, may be NULL: numCtus x 3 x 35 int32 (modeIdc, typeIdc, typeAuxInfo, offset[32]) as coded */
} hm355_sao_desc;
int hm355_sao_run(hm355_ctx *ctx, int n, hm355_sao_desc *descs);

/* ---- bitstream pass: TEncSlice::encodeSlice (TEncSlice.cpp:910-1095, called from TEncGOP.cpp:1559) on the pictures of slots 0..n-1 as
 * the search (hm355_run / hm355_compress_slices_inter) and, when sao_enabled is set, hm355_sao_run left them: the CABAC-coded slice data,
 * one substream per CTU row with WPP, else one.  Each substream ends with the terminating bin, the CABAC flush (TEncBinCABAC::finish)
 * and the byte alignment -- exactly the bytes TEncGOP.cpp:1572-1583 concatenates behind the slice header, before emulation prevention;
 * sub_sizes are the entry point sizes of the slice header (TEncSlice.cpp:1067-1071 adds the emulation count).  Slice header, parameter
 * sets and NAL packing stay with the caller. ---- */
typedef struct {
  int32_t slice_type, qp;              /* 2 = I, 1 = P, 0 = B; slice QP */
  int32_t cabac_init_type;             /* context table of a P / B slice (TEncSbac::resetEntropy :106-115); ignored for I */
  int32_t num_ref_idx[2], mvd_l1_zero, max_merge_cand;   /* slice header values the PU syntax depends on (P / B) */
  int32_t sao_enabled[2];              /* slice_sao_luma_flag, slice_sao_chroma_flag: both 0 = no SAO syntax (SPS SAO off, or switched off for the slice) */
  uint8_t *out; size_t out_cap;        /* out: the substreams back to back */
  uint32_t *sub_sizes;                 /* out: [hm355_num_substreams()] bytes of each substream */
  int32_t next_cabac_init_type;        /* out: TEncSbac::determineCabacInitIdx (:163-222): the table the PPS carries for the following pictures */
  uint32_t num_bins;                   /* out: bins coded (TEncBinCABAC::getBinsCoded) */
} hm355_bits_desc;
int hm355_num_substreams(const hm355_ctx *ctx);
int hm355_encode_slices_run(hm355_ctx *ctx, int n, hm355_bits_desc *descs);
/* host buffers in: the CTU data of one slice in the layout the search returns it (ictus NULL for an I slice), sao = numCtus x 3 x 35 int32 as
 * hm355_sao_run returns them (NULL with sao_enabled 0).  Uses slot 0. */
int hm355_encode_slice(hm355_ctx *ctx, hm355_bits_desc *desc, const hm355_ctu_out *ctus, const hm355_ctu_inter_out *ictus, const int32_t *sao);

/* ---- picture ingest and output: TVideoIOYuv::read / ::write (TVideoIOYuv.cpp:633-792) for planar 4:2:0 files, as TAppEncTop::encode drives them
 * (:431 read into the padded source picture, :603 write of the reconstruction with the conformance window).  The frames travel as they are on
 * disk (8-bit samples, or 16-bit little endian when file_bit_depth > 8); bit-depth scaling, padding by repetition up to the configured
 * (padded) width x height, cropping, rounding and clipping happen on the device.
 * hm355_upload_file_frames: frames[i] -> original planes of slot i.  hm355_download_file_frames: the reconstruction (source 0) or the original
 * planes (source 1) of slot i -> frames[i] of (width - conf_right) x (height - conf_bottom) samples at file_bit_depth.
 * hm355_download_org: the original planes of a slot as the encoder sees them (the TComPicYuv view of the padded source picture). ---- */
int hm355_upload_file_frames(hm355_ctx *ctx, int n, const void *const *frames, int file_width, int file_height, int file_bit_depth);
int hm355_download_file_frames(hm355_ctx *ctx, int n, void *const *frames, int file_bit_depth, int conf_right, int conf_bottom, int source);
int hm355_download_org(hm355_ctx *ctx, int slot, hm355_planes *org);

/* ---- cu_qp_delta: the hooks adaptive QP and rate control have inside compressSlice (TEncSlice.cpp:767-808 hands the CTU its QP through
 * TEncRateCtrl::setRCQP, TEncCu::xComputeQP :1154 derives it from the TEncPreanalyzer activities; TEncCu::xCheckDQP :1742, the split candidate
 * :1052-1085 and TEncEntropy.cpp:343-351 price / code the delta QP; TComDataCU::m_phQP feeds the deblocking filter).  MaxCuDQPDepth 0 (the CTU
 * is the quantisation group), MaxDeltaQP 0.  hm355_set_dqp arms a slot: its following searches (hm355_run*, hm355_compress_slice(s)(_inter)),
 * hm355_deblock_run and hm355_encode_slices_run run with cu_qp_delta enabled until it is called again with NULL / use_dqp 0.
 * hm355_get_dqp returns what the search left: m_phQP of every CTU ([numCtus*256]) and TEncCu::m_bEncodeDQP (input of the next picture).
 * hm355_preanalyze: TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64) for layer 0 on the slot's original picture -- the integer part on
 * the device: sums[a*8 + 0..3] = sum, [4..7] = sum of squares of the luma samples of CTU a's four quadrants; the caller derives the activities
 * and the QP offsets in double precision as the reference does (hm-16.2_amd/host/TEncTop.cpp, hm-16.2_amd/hm355.py aq_ctu_qp). ---- */
typedef struct {
  int32_t use_dqp;                     /* PPS cu_qp_delta_enabled_flag */
  int32_t dqp_flag_in;                 /* TEncCu::m_bEncodeDQP on entry (what the previous picture's encodeSlice left) */
  const int8_t *ctu_qp;                /* [numCtus] QP of every CTU; NULL: the slice QP (picture-level rate control) */
} hm355_dqp_desc;
int hm355_set_dqp(hm355_ctx *ctx, int slot, const hm355_dqp_desc *desc);
int hm355_get_dqp(hm355_ctx *ctx, int slot, int8_t *qp_out, int32_t *dqp_flag_out);
int hm355_preanalyze(hm355_ctx *ctx, int slot, uint64_t *sums);

/* ---- device-resident variant (what bench.py times: inputs already in HBM) ----
 * Upload / run / download are separate so that a caller can keep pictures resident. */
int hm355_upload(hm355_ctx *ctx, int slot, const hm355_planes *org);          /* host -> HBM picture slot */
int hm355_run(hm355_ctx *ctx, int n, const hm355_slice_desc *slices);         /* slots [0,n) -> results in HBM; blocking */
int hm355_download(hm355_ctx *ctx, int slot, hm355_planes *rec, hm355_ctu_out *ctus, hm355_slice_stats *stats);
/* Pipelined steps (the calling pattern of TEncGOP::compressGOP -> compressSlice, TEncGOP.cpp:1138, for a caller that keeps several groups
 * of independent pictures in flight): hm355_run_begin enqueues the search over slots [first_slot, first_slot + n) on pipeline lane
 * `lane` (0..3) and returns; hm355_run_wait blocks until that lane's launch has finished and reports its kernel time (HIP events on the
 * lane's stream) and errors.  Launches of different lanes run concurrently, each with its own stream and scratch areas, so the
 * wavefront drain of one group overlaps the fill of the next; slot ranges of launches in flight must not overlap.  I slices. */
int hm355_run_begin(hm355_ctx *ctx, int lane, int first_slot, int n, const hm355_slice_desc *slices);
/* how many launches the caller keeps in flight (1..4, default 1): each then takes that share of the device's resident searches, so that they
 * run side by side instead of one after the other (a waiting persistent workgroup keeps its place on the CU) */
int hm355_set_lane_share(hm355_ctx *ctx, int launches_in_flight);
int hm355_run_wait(hm355_ctx *ctx, int lane, double *kernel_ms);

/* CTU-row bands (TEncSlice.cpp:740-755,855-858 are the WPP hand-off points a band boundary cuts through): a picture is
 * searched by several devices, each owning a band of whole CTU rows [first_row, last_row] of the pictures in slots
 * [first_slot, first_slot + n) (WaveFrontSynchro=1, I slices).  The rows above first_row must be complete in the slot: searched by an
 * earlier hm355_run_rows on this context, or -- the row right above -- brought in with hm355_import_boundary from the device that owns it.
 * hm355_export_boundary copies out what the band below reads from CTU row `row`: the bottom sample line of the three planes (intra
 * reference samples, TComPattern.cpp:107-165), the CTUs' decision arrays (split-flag contexts, TComDataCU.cpp:1587) and the CABAC
 * state after each CTU (the WPP synchronisation source, TEncSlice.cpp:855-858); hm355_boundary_bytes() bytes per picture.  The buffers
 * may be host or device memory (the copies are address-space agnostic): a device buffer is handed to RCCL send / recv as it is, so a
 * boundary row never touches the host (bench.py --shard rows, hm-16.2_amd/bands.py).  The transport between devices is the caller's. */
int hm355_run_rows(hm355_ctx *ctx, int first_slot, int n, const hm355_slice_desc *slices, int first_row, int last_row);
size_t hm355_boundary_bytes(const hm355_ctx *ctx);
int hm355_export_boundary(hm355_ctx *ctx, int slot, int row, void *buf);
int hm355_import_boundary(hm355_ctx *ctx, int slot, int row, const void *buf);
/* kernel time of the last hm355_run in milliseconds, measured with HIP events on the launch stream,
 * and the number of kernel launches it took */
int hm355_last_run_info(const hm355_ctx *ctx, double *kernel_ms, int *launches);

/* ---- distortion / transform primitives as batched kernels (TComRdCost.cpp:465-1606,
 *      TComTrQuant.cpp:836-935); used by the known-answer parity tests and micro-benchmarks ----
 * blocks are n x n, tightly packed, `count` of them back to back. kind: 0 SAD, 1 SSE, 2 SATD */
int hm355_dist_batch(hm355_ctx *ctx, int kind, int n, int bit_depth, int count,
                     const int16_t *org, const int16_t *cur, uint32_t *out);
int hm355_transform_batch(hm355_ctx *ctx, int inverse, int n, int bit_depth, int use_dst, int count,
                          const int32_t *in, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif
