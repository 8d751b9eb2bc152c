// hm355 -- CTU rate-distortion search of an HM-16.2-shaped intra encoder, written for one 64-lane
// CDNA4 wavefront per CTU.
//
// Execution model of this file
//   * One workgroup == one wavefront (64 lanes) == one CTU in flight.  Control flow is wave-uniform:
//     every lane runs the decision logic on identical values (kept in LDS / HBM scratch), so there is
//     no divergence and no lane-0 bottleneck on branches.
//   * Sample / coefficient work (reference-sample fetch, the 35 predictors, residual, 4..32-point
//     separable transforms, de-quantisation, reconstruction, SAD/SSE/SATD) is spread over the lanes
//     with HM_PAR_FOR and finished by a DPP/shuffle butterfly reduction (hm_wave_sum).
//   * Transform blocks are staged through LDS with a padded stride (33 words) so that the row pass
//     and the column pass are both bank-conflict free; the transform matrix sits in LDS too.
//   * The CABAC estimator state (163 context bytes + Q15 accumulator) and its 5x6 snapshots stay in
//     LDS; snapshot copies are 22 eight-byte LDS moves.
// The file is plain C++: hm355_kernels.hip compiles it for gfx950; tests/hostsim compiles the very
// same source for the host with HM_NT == 1 (a debugging aid only -- it is not part of the product
// library and nothing in the product can reach it).
//
// What it restates (file:line under /root/reference/source/Lib): see each function.
#pragma once
#include <stddef.h>
#include "hm355_types.h"

#ifdef HM355_HOSTSIM
#include <string.h>
#include <math.h>
#include <stdlib.h>
#define HM_DEV static
#define HM_NOINLINE __attribute__((noinline))
#define HM_CONST static const
#define HM_ASSUME_LDS(p) ((void)0)
#define HM_ASSUME_GLB(p) ((void)0)
#define HM_NT 1
static inline int hm_lane() { return 0; }
#define HM_SYNC() ((void)0)
static inline uint32_t hm_wave_sum(uint32_t v) { return v; }
static inline int hm_wave_sum_i(int v) { return v; }
static inline int hm_wave_max_i(int v) { return v; }
#define HM_UNI(x) (x)
#define HM_UCALL(x) (x)
#define hm_uni_ptr(p) (p)
#define hm_uni_struct(v) (v)
#define HM_ENTRY(e) ((void)0)
#define HM_LDS_ADD(p, v) (*(p) += (v))
#ifdef HM355_HOSTSIM_REVERSE   /* run every lane-parallel loop backwards: catches order dependence */
#define HM_PAR_FOR_XY(x, y, w, n) for (int i_ = (n) - 1, y = i_ / (w), x = i_ - y * (w); i_ >= 0; i_--, x--, (x < 0 ? (x = (w) - 1, y--) : 0))
#define HM_PAR_FOR(i, n) for (int i = (n) - 1; i >= 0; i--)
#define HM_WAVE_FOR(k) for (int k = 63; k >= 0; k--)
#else
#define HM_PAR_FOR_XY(x, y, w, n) for (int i_ = 0, y = 0, x = 0; i_ < (n); i_++, x++, (x >= (w) ? (x = 0, y++) : 0))
#define HM_PAR_FOR(i, n) for (int i = 0; i < (n); i++)
#define HM_WAVE_FOR(k) for (int k = 0; k < 64; k++)
#endif
// lane variables (one value per lane of the wavefront) are plain arrays in the host twin
#define HM_LV(T, name) T name[64]
#define HM_LVARG(T, name) const T *name
#define HM_LVK(name, k) name[k]
#define HM_LV_GET(name, i) (name[i])
#define HM_LV_GETD(name, i) (name[i])
#define HM_LV_SET(name, i, v) (name[i] = (v))
#define HM_LV_SETD(name, i, v) (name[i] = (v))
#define HM_LV_GATHER(name, idx) (name[idx])
#define HM_BALLOT(m, k, cond) do { if (cond) (m) |= 1ull << (k); } while (0)
#define HM_ORDERED_ADD16(acc, name) do { for (int k_ = 15; k_ >= 0; k_--) (acc) += name[k_]; } while (0)
#define HM_ORDERED_CHAIN16(acc, a, b, pre) do { for (int k_ = 15; k_ >= 0; k_--) { pre[k_] = (acc); (acc) -= a[k_]; (acc) += b[k_]; } } while (0)
#else
#define HM_DEV __device__
#define HM_NOINLINE __attribute__((noinline))
#define HM_CONST __device__ const
/// address-space facts for pointers that arrive as generic function arguments / loaded values: lets the compiler
// emit ds_* / global_* instead of flat_* accesses
#if defined(__HIP_DEVICE_COMPILE__)
#define HM_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p)))
#define HM_ASSUME_GLB(p) __builtin_assume(!__builtin_amdgcn_is_shared((const void *)(p)) && !__builtin_amdgcn_is_private((const void *)(p)))
#else
#define HM_ASSUME_LDS(p) ((void)0)
#define HM_ASSUME_GLB(p) ((void)0)
#endif
#define HM_NT 64
__device__ __forceinline__ int hm_lane() { return (int)(threadIdx.x & 63u); }   // a workgroup is one wavefront, or a team of wavefronts (hm355_team.h)
// one wavefront per CTU: lanes run in lockstep, so a phase boundary only has to order this wave's own LDS /
// global accesses (wavefront-scope fence) -- no s_barrier and no drain of outstanding stores
#define HM_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// Wave reductions on the DPP path: four row_shr steps leave each 16-lane row's result in its last lane, the four row
// results are read with v_readlane and combined on the scalar unit.  The result is wave-uniform by construction
// (SGPR), which keeps the decision code that depends on it scalar.
template <int CTRL> __device__ __forceinline__ int hm_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int hm_wave_sum_i(int v)
{
  v += hm_dpp<0x111>(0, v); v += hm_dpp<0x112>(0, v); v += hm_dpp<0x114>(0, v); v += hm_dpp<0x118>(0, v);   // row_shr:1,2,4,8
  return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t hm_wave_sum(uint32_t v) { return (uint32_t)hm_wave_sum_i((int)v); }
__device__ __forceinline__ int hm_wave_max_i(int v)
{
  int t;
  t = hm_dpp<0x111>(v, v); v = t > v ? t : v; t = hm_dpp<0x112>(v, v); v = t > v ? t : v;
  t = hm_dpp<0x114>(v, v); v = t > v ? t : v; t = hm_dpp<0x118>(v, v); v = t > v ? t : v;
  const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31), c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
#define HM_LDS_ADD(p, v) atomicAdd((p), (v))
// wave-uniform value loaded through the vector path -> SGPR, so that the dependent control code runs on the scalar unit
#define HM_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
// Arguments and results of non-inlined device functions travel in VGPRs and count as divergent for the compiler.
// Every such function re-states at its entry that they are wave-uniform (one v_readfirstlane per dword), so that
// the decision logic compiles to scalar (SALU / s_cbranch) code; HM_ENTRY does the same for the pointer to the wavefront's LDS state
// (a workgroup holds one such state, or one per wavefront of a team: hm355_team.h).
template <class T> __device__ __forceinline__ T *hm_uni_ptr(T *p)
{
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T hm_uni_struct(T v)
{
  static_assert(sizeof(T) % 4 == 0, "dword-sized aggregates only");
  int w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
#define HM_ENTRY(e) do { (e) = hm_uni_ptr(e); HM_ASSUME_LDS(e); } while (0)
#define HM_UCALL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define HM_PAR_FOR(i, n) for (int i = hm_lane(); i < (n); i += HM_NT)
// lane-parallel loop over the first n samples of a rectangle of width w in raster order, with the (x, y) of a lane's sample kept
// incrementally (one division per loop instead of one per sample; w need not be a power of two: AMP widths 12, 24, 48)
#define HM_PAR_FOR_XY(x, y, w, n) \
  for (int i_ = hm_lane(), sy_ = HM_NT / (w), sx_ = HM_NT - sy_ * (w), y = i_ / (w), x = i_ - y * (w); i_ < (n); \
       i_ += HM_NT, x += sx_, y += sy_, (x >= (w) ? (x -= (w), y++) : 0))
// Lane variables: one value per lane of the wavefront, held in a VGPR.  HM_WAVE_FOR runs its body once on every
// lane (k = lane id, all lanes active); a wave-uniform lane index reads a lane with v_readlane and writes one with
// a compare + select (no memory access); a per-lane index gathers through ds_bpermute; HM_BALLOT collects a predicate.
#define HM_WAVE_FOR(k) for (int k = hm_lane(), k##_once = 1; k##_once; k##_once = 0)
#define HM_LV(T, name) T name
#define HM_LVARG(T, name) const T name
#define HM_LVK(name, k) name
#define HM_LV_GET(name, i) __builtin_amdgcn_readlane((int)(name), (int)(i))
#define HM_LV_GETD(name, i) hm_readlane_d((name), (int)(i))
#define HM_LV_SET(name, i, v) ((name) = (hm_lane() == (int)(i)) ? (v) : (name))        /* v_cmp + v_cndmask: no writelane builtin here */
#define HM_LV_SETD(name, i, v) ((name) = (hm_lane() == (int)(i)) ? (v) : (name))
#define HM_LV_GATHER(name, idx) __builtin_amdgcn_ds_bpermute((int)(idx) << 2, (int)(name))
#define HM_BALLOT(m, k, cond) ((m) = __ballot(cond))
// acc += lane 15, then lane 14, ... lane 0 of a double lane variable: the reference's position-by-position running sum
// (fp64 additions do not commute), fully unrolled so that every v_readlane has an immediate lane index
#define HM_ORDERED_ADD16(acc, name) do { _Pragma("unroll") for (int k_ = 15; k_ >= 0; k_--) (acc) += hm_readlane_d((name), k_); } while (0)
// the same for acc = (acc - a[k]) + b[k], remembering in pre[k] the value acc had before position k
#define HM_ORDERED_CHAIN16(acc, a, b, pre) do { _Pragma("unroll") for (int k_ = 15; k_ >= 0; k_--) { \
    (pre) = (hm_lane() == k_) ? (acc) : (pre); (acc) -= hm_readlane_d((a), k_); (acc) += hm_readlane_d((b), k_); } } while (0)
__device__ __forceinline__ double hm_readlane_d(double v, int i)
{ return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), i), __builtin_amdgcn_readlane(__double2loint(v), i)); }
#endif

// optional in-kernel cycle accounting (diagnostic build only: -DHM355_PROFILE, never in the product build)
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
#define HM_PROF_N 44
#define HM_PROF_BEGIN(e, id) const unsigned long long prof_t0_##id = __builtin_readcyclecounter()
#define HM_PROF_END(e, id) do { (e)->prof[id] += __builtin_readcyclecounter() - prof_t0_##id; (e)->profCnt[id] += 1; } while (0)
#else
#define HM_PROF_BEGIN(e, id) ((void)0)
#define HM_PROF_END(e, id) ((void)0)
#endif
enum { PR_RDOQ = 0, PR_BITS, PR_ADI, PR_PRED, PR_FWD, PR_INV, PR_SATD35, PR_TUBLK, PR_SAVE, PR_CHROMA, PR_LUMA, PR_ENCCU, PR_TOTAL,
       PR_S4L = 32, PR_S4C, PR_D0, PR_D1, PR_D2, PR_D3, PR_NXN, PR_S8L, PR_S4LEAF, PR_S4CLEAF, PR_S8C,
       PR_ME_INT = 16, PR_ME_FRAC, PR_AMVP, PR_MRG_EST, PR_MC, PR_IRQ, PR_IRES, PR_MRG2N, PR_INTERCU, PR_INTRA_IN_P, PR_IQ_FULL, PR_IQ_FWD, PR_IQ_RDOQ, PR_IQ_BITS, PR_IQ_INV, PR_IQ_ENC };

#define HM_MAX_DOUBLE 1.7e+308
#define PLANAR_IDX 0
#define DC_IDX 1
#define HOR_IDX 10
#define VER_IDX 26
#define DM_CHROMA_IDX 36
#define SIZE_2Nx2N 0
#define SIZE_NxN 3
#define SIZE_NONE 8
#define MODE_INTRA 1
#define MODE_NONE 2
#define SCAN_DIAG 0
#define SCAN_HOR 1
#define SCAN_VER 2

// ------------------------------------------------------------------------------------------------
// constant tables
// ------------------------------------------------------------------------------------------------
// ContextModel.cpp:66-128 (FAST_BIT_EST)
// (the MPS transition is arithmetic: hm_next_state)
HM_CONST uint8_t HM_NEXT_LPS[128] __attribute__((aligned(4))) = {
  1, 0, 0, 1, 2, 3, 4, 5, 4, 5, 8, 9, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 18, 19, 22, 23, 22, 23, 24, 25,
  26, 27, 26, 27, 30, 31, 30, 31, 32, 33, 32, 33, 36, 37, 36, 37, 38, 39, 38, 39, 42, 43, 42, 43, 44, 45, 44, 45, 46, 47, 48, 49,
  48, 49, 50, 51, 52, 53, 52, 53, 54, 55, 54, 55, 56, 57, 58, 59, 58, 59, 60, 61, 60, 61, 60, 61, 62, 63, 64, 65, 64, 65, 66, 67,
  66, 67, 66, 67, 68, 69, 68, 69, 70, 71, 70, 71, 70, 71, 72, 73, 72, 73, 72, 73, 74, 75, 74, 75, 74, 75, 76, 77, 76, 77, 126, 127 };
HM_CONST int32_t HM_ENTROPY_BITS[128] = {
  0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
  0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
  0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
  0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
  0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
  0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
  0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
  0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };
// I-slice context initialisation values, ContextTables.h:170-502 (our context order, see hm355_types.h)
HM_CONST uint8_t HM_CTX_INIT_I[HM_NUM_CTX] = {
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
// P- and B-slice rows of the same tables
HM_CONST uint8_t HM_CTX_INIT_P[HM_NUM_CTX] = {
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
HM_CONST uint8_t HM_CTX_INIT_B[HM_NUM_CTX] = {
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
// first column of the 32-point core transform (TComRom.cpp:456-484); the matrix follows the cosine index law
HM_CONST int8_t HM_DCT_C[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
HM_CONST int8_t HM_DST4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
HM_CONST int32_t HM_QUANT_SCALES[6] = {26214, 23302, 20560, 18396, 16384, 14564};
HM_CONST int32_t HM_INV_QUANT_SCALES[6] = {40, 45, 51, 57, 64, 72};
HM_CONST uint8_t HM_GROUP_IDX[32] = {0,1,2,3,4,4,5,5,6,6,6,6,7,7,7,7,8,8,8,8,8,8,8,8,9,9,9,9,9,9,9,9};
HM_CONST uint8_t HM_CTX_IND_MAP_4x4[16] = {0,1,4,5, 2,3,4,5, 6,6,8,8, 7,7,8,8};
HM_CONST uint8_t HM_INTRA_MODE_NUM_FAST[6] = {3, 8, 8, 3, 3, 3};
HM_CONST uint8_t HM_INTRA_FILTER[5] = {10, 7, 1, 0, 10};
HM_CONST int8_t HM_ANG_TABLE[9] = {0, 2, 5, 9, 13, 17, 21, 26, 32};
HM_CONST int16_t HM_INV_ANG_TABLE[9] = {0, 4096, 1638, 910, 630, 482, 390, 315, 256};

// ------------------------------------------------------------------------------------------------
// LDS-resident state of one CTU search
// ------------------------------------------------------------------------------------------------
// TU descriptor: the subset of TComTU (TComTU.h/.cpp) that 4:2:0 intra coding needs
struct TU {                            // 24 bytes: it sits in every frame of the LDS tree-walk stacks and travels by value (6 dwords)
  int16_t cuParts, parts, cParts, cOff;
  uint8_t cuZ, cuDepth, relZ, trDepth, log2, section, x, y;
  uint8_t cW, cCodeAll, cTrDepth, cRelZ, cx, cy, pad_[2];
};
static_assert(sizeof(TU) == 24, "TU layout");
// explicit stacks of the tree walks (kept in LDS: no private-memory traffic, no device recursion)
struct TuWalk { TU node[5]; int8_t next[5]; int sp; };
struct RqtFrame {
  TU t; int8_t phase, child, checkFull, checkSplit, bestModeId; uint32_t singleDist, singleCbf, splitDist, splitCbf, singleBits; double singleCost, splitCost;
};
struct CuFrame { int16_t cuZ; int8_t phase, sub, boundary, parentPart, ampSens, pad_; double bestCost, splitCost; uint32_t bestBits, bestDist, splitBits, splitDist; };
// inter (P / B slice) helpers kept in LDS
struct MvFieldD { MvD mv; int ref; };
struct MergeList { MvFieldD f[5][2]; uint8_t dir[5]; int num; };
struct AmvpInfo { MvD cand[3]; int n; };
struct TZ {                            // state of one integer motion search (IntTZSearchStruct, TEncSearch.h:121-133)
  const Pel *org; int orgStride, w, h;
  const Pel *ref; int refStride;
  uint32_t bestSad; int bestX, bestY, bestDist, bestRound, pointNr;
  int16_t l, r, t, b;                  // search range in integer samples (within +-64 of a clipped vector: 16 bits are plenty)
  int16_t subShift, pad_;
  int16_t lx[16], ly[16]; int8_t lp[16], ld[16];   // search points queued for one batched evaluation (x, y, point number, distance)
  int16_t wx0, wy0, ww, wh;            // reference window staged in LDS (tz_stage_window): origin in search-point coordinates, size in samples; ww == 0: none
};
struct IrqFrame {                      // one level of the inter residual quadtree (xEstimateResidualQT)
  TU t; int8_t phase, child, checkFull, checkSplit, zero; uint8_t bestTS[3];
  uint32_t absSum[3], bestCBF[3], singleBits, singleDist, subBits, subDist; double singleCost, subCost;
};

#define HM_TSTRIDE 33
#define HM_RQ_LDS 256                  // RDOQ per-position arrays live in LDS up to 16x16, in HBM scratch for 32x32
struct RqLds {                         // indexed by scan position
  int32_t lvl[HM_RQ_LDS];              // |coef| * quant scale (lLevelDouble)
  uint16_t pos[HM_RQ_LDS];             // raster position | sign << 15
  uint16_t dec[HM_RQ_LDS];             // level at decision time
  int16_t cur[HM_RQ_LDS];              // working / final level
  uint8_t ctxSig[HM_RQ_LDS];           // significance context
  uint8_t code[HM_RQ_LDS];             // significance cost of the position: 0 none, 1 bits(ctx,0), 2 bits(ctx,1)
  // per coefficient group (all block sizes)
  uint8_t cgCtxSet[64];                // context set each coefficient group started with
  uint8_t cgFlag[64];
};
struct RefLds {
  Pel refTop[2][132], refLeft[2][132]; // [filtered][0 = corner, 1..2N]
  Pel refMain[200], refSide[200];      // angular: extended main / side reference, origin at +64
  Pel line[272];
};
// slot index: CI_CURR_BEST / CI_NEXT_BEST are never addressed at depth 4
#define HM_SLOT(d, ci) ((d) * CI_NUM + (ci) - ((d) == 4 ? 2 : 0))
#define HM_NUM_SLOTS (4 * CI_NUM + 3)
struct Team;                           // hm355_team.h: the wavefronts of one workgroup searching one CTU together (latency mode)
#define HM_CTU_WAVES 12                 /* independent CTU searches (wavefronts) per workgroup of hm355_ctu_kernel: they share one LdsTables */
// read-only tables every CTU search of a workgroup shares (one copy per workgroup: the searches of a workgroup are independent wavefronts)
struct LdsTables {
  int8_t tmat[32 * HM_TSTRIDE];        // 32-point transform matrix, padded rows
  int8_t dst4[16];                     // the 4-point DST of intra luma 4x4 blocks
  int32_t ebits[128];                  // HM_ENTROPY_BITS for per-lane lookups (hm355_simt4.h)
};
#if defined(HM355_HOSTSIM)
static LdsTables g_lt;
#else
__shared__ LdsTables g_lt;
#endif
#define HM_LT() (&g_lt)
struct Shared {
  Cabac cur;                           // m_pcRDGoOnSbacCoder
  int32_t bufA[32 * HM_TSTRIDE];
  union {                              // phase-exclusive LDS: transform temp | RDOQ state | intra reference samples
    int32_t bufB[32 * HM_TSTRIDE];
    RqLds rq;
    RefLds ref;
  } u;
  TuWalk walkOuter, walkInner;         // tree-walk stacks
  CuFrame cuf[4];
  // results handed back by the big non-inlined stages (instead of pointers to private memory)
  double outCost; uint32_t outBits, outDist; double outRdCost;
  int32_t mpmZ;                        // the PU whose most-probable-mode list is tabulated below (-1: none)
  int32_t s8Winner, s8Reuse;           // hm355_simt8.h: the first pass's winning candidate; set while its evaluation can stand in for the closing pass's unsplit 8x8 TU
  // inter (P / B slice) state that outlives a stage
  InterMeta *im;                       // motion arrays of the CTU under search (HBM)
  uint32_t mcost; MvD mvPredictor; int32_t costScale;   // TComRdCost motion-cost state
  // Scratch state of the intra stages and of the inter stages share their LDS: nothing of the one is live while the other runs (an intra
  // trial of a P / B slice runs between inter trials, never inside one; the host twin runs the same layout against the P / B fixtures)
  union {
    struct {                           // est_intra_pred_qt / recur_intra_coding_qt / chroma / init_adi_pattern
      uint8_t flags[72];
      RqtFrame rqt[4]; uint32_t rqtRetDist[5], rqtRetBits[5]; double rqtRetCost[5];
      int32_t rdModeList[12];
      uint8_t splitCbf[5][2];
      uint32_t outDistY;
      uint32_t satd[36];               // SATD of the 35 intra modes of the PU under test
      int32_t mpmNum, mpmPreds[3];     // most-probable-mode list of the PU under test (same for all its candidates)
    };
    struct {                           // motion search, merge / AMVP lists, inter residual quadtree
      int32_t absCoeff[10];            // the 9 costs of a fractional refinement step
      TZ tz; MvD outMv; MergeList ml; AmvpInfo amvp; MvFieldD mrgField[2]; int32_t mrgDir, mrgIdx; uint32_t mrgCost, irqZeroDist;
      IrqFrame irq[4];
    };
  };
  // uniform per-CTU context
  int32_t width, height, bitDepth, wCtu, stride[3];
  const Params *P; FrameBuf fb; WorkSpace *ws; const Tables *tab;
  CtuMeta meta;                        // decision arrays of the CTU under search (written back to HBM at the end)
  TCoeff *cc;
  int32_t ctuX, ctuY, ctuAddr;
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  unsigned long long prof[HM_PROF_N]; unsigned long long profCnt[HM_PROF_N];
#endif
};

// optional RD-evaluation trace (diagnostic builds only: -DHM355_TRACE; the host twin prints, the device appends to P->prof)
#if defined(HM355_TRACE) && defined(HM355_HOSTSIM)
#include <stdio.h>
static FILE *g_hm_trace;
#define HM_TRACE(e, tag, a, b, c) do { if (g_hm_trace) fprintf(g_hm_trace, "%d %d %u %u %.3f\n", (int)(tag), (e)->ctuAddr, (unsigned)(a), (unsigned)(b), (double)(c)); } while (0)
#elif defined(HM355_TRACE)
#define HM_TRACE_CAP (1u << 21)
__device__ inline void hm_trace(Shared *e, int tag, uint32_t a, uint32_t b, double c)
{
  if (hm_lane() == 0) {
    unsigned long long *t = e->P->prof; const unsigned long long n = atomicAdd(t, 1ull);
    if (n < HM_TRACE_CAP) { t[1 + n * 3] = ((unsigned long long)(uint32_t)tag << 32) | (uint32_t)e->ctuAddr; t[2 + n * 3] = ((unsigned long long)a << 32) | b; t[3 + n * 3] = (unsigned long long)__double_as_longlong(c); }
  }
}
#define HM_TRACE(e, tag, a, b, c) hm_trace((e), (tag), (uint32_t)(a), (uint32_t)(b), (double)(c))
#else
#define HM_TRACE(e, tag, a, b, c) ((void)0)
#endif

#ifndef HM355_HOSTSIM
__shared__ Shared g_sh;                // the one wavefront of the secondary kernels' workgroups
__shared__ Shared g_shs[HM_CTU_WAVES];  // the CTU searches of a workgroup of hm355_ctu_kernel
#endif
#if !defined(HM355_PROFILE)
static_assert(HM_CTU_WAVES * (sizeof(Shared) + 16) + sizeof(LdsTables) <= 163840 / (12 / HM_CTU_WAVES) && sizeof(Shared) % 8 == 0, "two workgroups of HM_CTU_WAVES searches (+ work items, + the shared tables) per CU: 12 CTU searches in 160 KB of LDS");
#endif
static_assert(offsetof(Shared, bufA) % 8 == 0 && (16 * HM_TSTRIDE * 4) % 8 == 0, "the RDOQ cost array aliases the lower half of bufA as doubles");
static_assert(offsetof(Shared, u) == offsetof(Shared, bufA) + sizeof(((Shared *)0)->bufA), "the motion search stages its reference window across bufA and the union behind it");
#define HM_TZ_WIN_SAMPLES ((int)((sizeof(((Shared *)0)->bufA) + sizeof(((Shared *)0)->u)) / sizeof(Pel)))

HM_DEV inline int hm_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
HM_DEV inline int hm_abs(int v) { return v < 0 ? -v : v; }
// z-scan <-> raster order of the 256 4x4 partitions of a CTU (g_auiZscanToRaster / g_auiRasterToZscan, TComRom.cpp:52-86):
// z interleaves the bits of (y4, x4), x in the even positions
HM_DEV inline int hm_z2r(int z)
{
  const int x = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4) | ((z >> 3) & 8);
  const int y = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4) | ((z >> 4) & 8);
  return (y << 4) | x;
}
HM_DEV inline int hm_spread4(int v) { return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2) | ((v & 8) << 3); }
HM_DEV inline int hm_r2z(int r) { return hm_spread4(r & 15) | (hm_spread4(r >> 4) << 1); }
HM_DEV inline int hm_log2(int n) { return n >= 64 ? 6 : (n >= 32 ? 5 : (n >= 16 ? 4 : (n >= 8 ? 3 : 2))); }

// lane-parallel byte fill / copies (uniform arguments)
HM_DEV inline void par_set8(uint8_t *p, int v, int n) { HM_PAR_FOR(i, n) p[i] = (uint8_t)v; HM_SYNC(); }
HM_DEV inline void par_copy8(uint8_t *d, const uint8_t *s, int n) { HM_PAR_FOR(i, n) d[i] = s[i]; HM_SYNC(); }
HM_DEV inline void par_copy32(TCoeff *d, const TCoeff *s, int n) { HM_PAR_FOR(i, n) d[i] = s[i]; HM_SYNC(); }
HM_DEV inline void par_zero32(TCoeff *d, int n) { HM_PAR_FOR(i, n) d[i] = 0; HM_SYNC(); }
// n x n block copy between two strided planes
HM_DEV inline void par_copy_blk(Pel *d, int ds, const Pel *s, int ss, int n)
{
  const int l2 = hm_log2(n);
  HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); d[y * ds + x] = s[y * ss + x]; }
  HM_SYNC();
}

// ------------------------------------------------------------------------------------------------
// CABAC bit estimator (TEncBinCABACCounter, TEncBinCoderCABACCounter.cpp:56-131)
// ------------------------------------------------------------------------------------------------
HM_DEV inline void cabac_copy(Cabac *d, const Cabac *s)
{ // TEncSbac::load/store, TEncSbac.cpp:396-421: the whole context array + m_fracBits
  const uint64_t *sp = (const uint64_t *)s; uint64_t *dp = (uint64_t *)d;
  HM_PAR_FOR(i, (int)(sizeof(Cabac) / 8)) dp[i] = sp[i];
  HM_SYNC();
}
// A snapshot held in registers for the duration of one function (one uint64 per lane): restoring it is an LDS write
// instead of a round trip to the HBM snapshot slots.  The host twin keeps a plain copy.
#if defined(HM355_HOSTSIM)
typedef Cabac CabacHold;
static inline CabacHold cabac_hold(const Cabac *s) { return *s; }
static inline void cabac_put(Cabac *d, const CabacHold &h) { *d = h; }
#else
typedef uint64_t CabacHold;
__device__ __forceinline__ CabacHold cabac_hold(const Cabac *s)
{ const int l = hm_lane(); return l < (int)(sizeof(Cabac) / 8) ? ((const uint64_t *)s)[l] : 0ull; }
__device__ __forceinline__ void cabac_put(Cabac *d, CabacHold h)
{ const int l = hm_lane(); if (l < (int)(sizeof(Cabac) / 8)) ((uint64_t *)d)[l] = h; HM_SYNC(); }
#endif
HM_DEV inline void cabac_init(Cabac *c, int qp, int initType = 2)
{ // ContextModel::init, ContextModel.cpp:55-64; TEncSbac::resetEntropy, TEncSbac.cpp:106-161
  qp = hm_clip3(0, 51, qp);
  HM_PAR_FOR(i, HM_NUM_CTX) {
    const int iv = initType == 2 ? HM_CTX_INIT_I[i] : (initType == 1 ? HM_CTX_INIT_P[i] : HM_CTX_INIT_B[i]);
    const int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
    const int mps = st >= 64;
    c->s[i] = (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
  }
  c->frac = 0;
  HM_SYNC();
}
// next context state (ContextModel::updateMPS/updateLPS): the MPS transition is +2 saturating at state 62
HM_DEV inline int hm_next_state(int st, int bin) { return bin == (st & 1) ? (st < 124 ? st + 2 : st) : HM_NEXT_LPS[st]; }
HM_DEV inline void enc_bin(const Shared *e, Cabac *c, int ctx, int bin)
{
  const int st = c->s[ctx];
  c->frac += (uint64_t)HM_LT()->ebits[st ^ bin];
  c->s[ctx] = hm_next_state(st, bin);
}
HM_DEV inline void enc_ep(Cabac *c, int n) { c->frac += (uint64_t)32768 * (uint64_t)n; }
HM_DEV inline void enc_trm(const Shared *e, Cabac *c, int bin) { c->frac += (uint64_t)HM_ENTROPY_BITS[126 ^ bin]; }
HM_DEV inline void reset_bits(Cabac *c) { c->frac &= 32767; }           // TEncBinCoderCABAC.cpp:161
HM_DEV inline uint32_t num_bits(const Cabac *c) { return (uint32_t)(c->frac >> 15); }

HM_DEV inline double calc_rd_cost(const Shared *e, uint32_t bits, uint32_t dist)
{ // TComRdCost::calcRdCost, TComRdCost.cpp:56-123 (DF_DEFAULT, lossy)
  const double t = (double)bits * e->fb.lambda;
  const double u = (double)dist + t;
  return floor(u + 0.5);
}

struct Rect;
template <class C> HM_DEV inline void code_skip_flag(Shared *e, C *c, int z);   // hm355_inter.h

// ------------------------------------------------------------------------------------------------
// distortion (TComRdCost.cpp): lanes split the samples / the Hadamard blocks, butterfly-reduce
// ------------------------------------------------------------------------------------------------
HM_DEV inline uint32_t dist_sse(const Pel *org, int so, const Pel *cur, int sc, int n, int bitDepth)
{ // xGetSSE*, TComRdCost.cpp:970-1318
  const int shift = (bitDepth - 8) << 1, l2 = hm_log2(n);
  uint32_t sum = 0;
  HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); const int d = org[y * so + x] - cur[y * sc + x]; sum += (uint32_t)((d * d) >> shift); }
  return hm_wave_sum(sum);
}
HM_DEV inline uint32_t had8(const Pel *org, int so, const Pel *cur, int sc)
{ // xCalcHADs8x8, TComRdCost.cpp:1439-1530: sum |H8 D H8|, (s+2)>>2
  int d[64];
#pragma unroll
  for (int y = 0; y < 8; y++)
#pragma unroll
    for (int x = 0; x < 8; x++) d[y * 8 + x] = org[y * so + x] - cur[y * sc + x];
#pragma unroll
  for (int y = 0; y < 8; y++) {
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
      for (int b = 0; b < 8; b += len << 1)
#pragma unroll
        for (int k = 0; k < len; k++) { const int a0 = d[y * 8 + b + k], a1 = d[y * 8 + b + k + len]; d[y * 8 + b + k] = a0 + a1; d[y * 8 + b + k + len] = a0 - a1; }
  }
  uint32_t s = 0;
#pragma unroll
  for (int x = 0; x < 8; x++) {
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
      for (int b = 0; b < 8; b += len << 1)
#pragma unroll
        for (int k = 0; k < len; k++) { const int a0 = d[(b + k) * 8 + x], a1 = d[(b + k + len) * 8 + x]; d[(b + k) * 8 + x] = a0 + a1; d[(b + k + len) * 8 + x] = a0 - a1; }
#pragma unroll
    for (int y = 0; y < 8; y++) s += (uint32_t)hm_abs(d[y * 8 + x]);
  }
  return (s + 2) >> 2;
}
HM_DEV inline uint32_t had4(const Pel *org, int so, const Pel *cur, int sc)
{ // xCalcHADs4x4, TComRdCost.cpp:1343-1437: sum |H4 D H4|, (s+1)>>1
  int d[16];
#pragma unroll
  for (int y = 0; y < 4; y++)
#pragma unroll
    for (int x = 0; x < 4; x++) d[y * 4 + x] = org[y * so + x] - cur[y * sc + x];
#pragma unroll
  for (int y = 0; y < 4; y++) {
    const int a = d[y * 4] + d[y * 4 + 1], b = d[y * 4] - d[y * 4 + 1], c = d[y * 4 + 2] + d[y * 4 + 3], e = d[y * 4 + 2] - d[y * 4 + 3];
    d[y * 4] = a + c; d[y * 4 + 1] = b + e; d[y * 4 + 2] = a - c; d[y * 4 + 3] = b - e;
  }
  uint32_t s = 0;
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int a = d[x] + d[4 + x], b = d[x] - d[4 + x], c = d[8 + x] + d[12 + x], e = d[8 + x] - d[12 + x];
    s += (uint32_t)(hm_abs(a + c) + hm_abs(b + e) + hm_abs(a - c) + hm_abs(b - e));
  }
  return (s + 1) >> 1;
}
HM_DEV inline uint32_t dist_hads(const Pel *org, int so, const Pel *cur, int sc, int n, int bitDepth)
{ // xGetHADs, TComRdCost.cpp:1537-1606
  uint32_t sum = 0; const int l2 = hm_log2(n);
  if (n >= 8) { const int nb = n >> 3; HM_PAR_FOR(b, nb * nb) { const int by = b >> (l2 - 3), bx = b & (nb - 1); sum += had8(org + by * 8 * so + bx * 8, so, cur + by * 8 * sc + bx * 8, sc); } }
  else { HM_PAR_FOR(b, 1) sum += had4(org, so, cur, sc); }
  return hm_wave_sum(sum) >> (bitDepth - 8);
}
HM_DEV inline uint32_t dist_sad(const Pel *org, int so, const Pel *cur, int sc, int n, int subShift, int bitDepth)
{ // xGetSAD*, TComRdCost.cpp:465-962
  const int l2 = hm_log2(n), rows = n >> subShift;
  uint32_t sum = 0;
  HM_PAR_FOR(i, rows * n) { const int y = (i >> l2) << subShift, x = i & (n - 1); sum += (uint32_t)hm_abs(org[y * so + x] - cur[y * sc + x]); }
  return (hm_wave_sum(sum) << subShift) >> (bitDepth - 8);
}

// ------------------------------------------------------------------------------------------------
// transforms (TComTrQuant.cpp:387-935).  Blocks live in LDS with row stride HM_TSTRIDE.
// ------------------------------------------------------------------------------------------------
HM_DEV inline void load_tables()
{
  HM_PAR_FOR(i, 1024) {
    const int k = i >> 5, n = i & 31, m = (k * (2 * n + 1)) & 127;
    int v;
    if (m <= 32) v = HM_DCT_C[m]; else if (m <= 64) v = -HM_DCT_C[64 - m]; else if (m <= 96) v = -HM_DCT_C[m - 64]; else v = HM_DCT_C[128 - m];
    HM_LT()->tmat[k * HM_TSTRIDE + n] = (int8_t)v;
  }
  HM_PAR_FOR(i, 16) HM_LT()->dst4[i] = HM_DST4[i];
  HM_PAR_FOR(i, 128) HM_LT()->ebits[i] = HM_ENTROPY_BITS[i];
  HM_SYNC();
}
HM_DEV inline int tm(const Shared *e, int n, int dst, int k, int j) { return dst ? HM_LT()->dst4[k * 4 + j] : HM_LT()->tmat[(k * (32 / n)) * HM_TSTRIDE + j]; }

// The DCT rows are symmetric (even k) or antisymmetric (odd k) about their middle -- what partialButterfly4/8/16/32 (TComTrQuant.cpp:387-763) build
// on -- so an output needs n/2 products of the folded input (x[i] +- x[n-1-i]) instead of n.  One fold level: the sums are the same integers in
// another order, and half the multiplies and half the LDS reads go.  The 4-point DST has no such symmetry and takes the plain product.
HM_DEV inline void fold_rows(int32_t *X, int n, int l2)
{ // every row of X in place: [0, n/2) <- x[i] + x[n-1-i], [n/2, n) <- x[i] - x[n-1-i].  The pairs (i, n-1-i) and (n/2-1-i, n/2+i) read and write the
  // same four positions, so one lane takes both: in place whatever the order the lanes run in.
  const int h = n >> 1, q = n >> 2;
  HM_PAR_FOR(o, n * q) {
    const int j = o >> (l2 - 2), i = o & (q - 1);
    int32_t *r = X + j * HM_TSTRIDE;
    const int32_t a = r[i], b = r[n - 1 - i], c = r[h - 1 - i], d = r[h + i];
    r[i] = a + b; r[h + i] = a - b; r[h - 1 - i] = c + d; r[n - 1 - i] = c - d;
  }
  HM_SYNC();
}
// forward: src (bufA, [row][col]) -> dst (bufA); xTrMxN, TComTrQuant.cpp:836-890
HM_DEV HM_NOINLINE void fwd_transform(Shared *e, int n, int useDst, int bitDepth)
{
  HM_ENTRY(e); n = HM_UNI(n); useDst = HM_UNI(useDst); bitDepth = HM_UNI(bitDepth);
  const int l2 = hm_log2(n), s1 = l2 + bitDepth + 6 - 15, s2 = l2 + 6;
  const int a1 = s1 > 0 ? 1 << (s1 - 1) : 0, a2 = 1 << (s2 - 1);
  int32_t *A = e->bufA, *B = e->u.bufB;
  if (useDst) {
    HM_PAR_FOR(o, n * n) { // o = j*n + k with k fastest: lanes of one row share the source row (LDS broadcast)
      const int j = o >> l2, k = o & (n - 1);
      int32_t acc = 0;
      for (int i = 0; i < n; i++) acc += tm(e, n, 1, k, i) * A[j * HM_TSTRIDE + i];
      B[k * HM_TSTRIDE + j] = (acc + a1) >> s1;
    }
    HM_SYNC();
    HM_PAR_FOR(o, n * n) {
      const int j = o >> l2, k = o & (n - 1);
      int32_t acc = 0;
      for (int i = 0; i < n; i++) acc += tm(e, n, 1, k, i) * B[j * HM_TSTRIDE + i];
      A[k * HM_TSTRIDE + j] = (acc + a2) >> s2;
    }
    HM_SYNC();
    return;
  }
  const int h = n >> 1, step = 32 / n;
  fold_rows(A, n, l2);
  HM_PAR_FOR(o, n * n) {
    const int j = o >> l2, k = o & (n - 1);
    const int32_t *x = A + j * HM_TSTRIDE + ((k & 1) ? h : 0); const int8_t *t = HM_LT()->tmat + (k * step) * HM_TSTRIDE;
    int32_t acc = 0;
    for (int i = 0; i < h; i++) acc += t[i] * x[i];
    B[k * HM_TSTRIDE + j] = (acc + a1) >> s1;
  }
  HM_SYNC();
  fold_rows(B, n, l2);
  HM_PAR_FOR(o, n * n) {
    const int j = o >> l2, k = o & (n - 1);
    const int32_t *x = B + j * HM_TSTRIDE + ((k & 1) ? h : 0); const int8_t *t = HM_LT()->tmat + (k * step) * HM_TSTRIDE;
    int32_t acc = 0;
    for (int i = 0; i < h; i++) acc += t[i] * x[i];
    A[k * HM_TSTRIDE + j] = (acc + a2) >> s2;
  }
  HM_SYNC();
}
// inverse: coefficients in bufA -> residual in bufA; xITrMxN, TComTrQuant.cpp:894-935.  Outputs i and n-1-i share the even-row sum and differ in the
// sign of the odd-row sum: one lane makes both from n products.
HM_DEV HM_NOINLINE void inv_transform(Shared *e, int n, int useDst, int bitDepth)
{
  HM_ENTRY(e); n = HM_UNI(n); useDst = HM_UNI(useDst); bitDepth = HM_UNI(bitDepth);
  const int l2 = hm_log2(n), s1 = 7, s2 = 20 - bitDepth;
  int32_t *A = e->bufA, *B = e->u.bufB;
  if (useDst) {
    HM_PAR_FOR(o, n * n) {
      const int j = o >> l2, i = o & (n - 1);
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += tm(e, n, 1, k, i) * A[k * HM_TSTRIDE + j];
      B[j * HM_TSTRIDE + i] = hm_clip3(-32768, 32767, (acc + (1 << (s1 - 1))) >> s1);
    }
    HM_SYNC();
    HM_PAR_FOR(o, n * n) {
      const int j = o >> l2, i = o & (n - 1);
      int32_t acc = 0;
      for (int k = 0; k < n; k++) acc += tm(e, n, 1, k, i) * B[k * HM_TSTRIDE + j];
      A[j * HM_TSTRIDE + i] = hm_clip3(-32768, 32767, (acc + (1 << (s2 - 1))) >> s2);
    }
    HM_SYNC();
    return;
  }
  const int h = n >> 1, step = 32 / n;
  HM_PAR_FOR(o, n * h) {
    const int j = o >> (l2 - 1), i = o & (h - 1);
    int32_t ev = 0, od = 0;
    for (int k = 0; k < n; k += 2) { ev += HM_LT()->tmat[(k * step) * HM_TSTRIDE + i] * A[k * HM_TSTRIDE + j]; od += HM_LT()->tmat[((k + 1) * step) * HM_TSTRIDE + i] * A[(k + 1) * HM_TSTRIDE + j]; }
    B[j * HM_TSTRIDE + i] = hm_clip3(-32768, 32767, (ev + od + (1 << (s1 - 1))) >> s1);
    B[j * HM_TSTRIDE + n - 1 - i] = hm_clip3(-32768, 32767, (ev - od + (1 << (s1 - 1))) >> s1);
  }
  HM_SYNC();
  HM_PAR_FOR(o, n * h) {
    const int j = o >> (l2 - 1), i = o & (h - 1);
    int32_t ev = 0, od = 0;
    for (int k = 0; k < n; k += 2) { ev += HM_LT()->tmat[(k * step) * HM_TSTRIDE + i] * B[k * HM_TSTRIDE + j]; od += HM_LT()->tmat[((k + 1) * step) * HM_TSTRIDE + i] * B[(k + 1) * HM_TSTRIDE + j]; }
    A[j * HM_TSTRIDE + i] = hm_clip3(-32768, 32767, (ev + od + (1 << (s2 - 1))) >> s2);
    A[j * HM_TSTRIDE + n - 1 - i] = hm_clip3(-32768, 32767, (ev - od + (1 << (s2 - 1))) >> s2);
  }
  HM_SYNC();
}

// ------------------------------------------------------------------------------------------------
// TU descriptor: the subset of TComTU (TComTU.h/.cpp) that 4:2:0 intra coding needs
// ------------------------------------------------------------------------------------------------
HM_DEV inline TU tu_root(const Shared *e, int cuZ, int cuDepth)
{ // TComTU::TComTU(pcCU, absPartIdxCU, cuDepth, 0), TComTU.cpp:48
  TU t;
  t.cuZ = (int16_t)cuZ; t.cuDepth = (int16_t)cuDepth; t.cuParts = (int16_t)(256 >> (2 * cuDepth));
  t.relZ = 0; t.trDepth = 0; t.log2 = (int16_t)(6 - cuDepth); t.parts = t.cuParts; t.section = 0;
  const int r = hm_z2r(cuZ);
  t.x = (int16_t)((r & 15) * 4); t.y = (int16_t)((r >> 4) * 4);
  t.cW = (int16_t)(1 << (t.log2 - 1)); t.cCodeAll = 1; t.cTrDepth = 0; t.cRelZ = 0; t.cParts = t.parts; t.cOff = (int16_t)(cuZ * 4);
  t.cx = t.x >> 1; t.cy = t.y >> 1;
  return t;
}
HM_DEV inline TU tu_child(const TU *p, int section, int processLast)
{ // TComTU::TComTU(parent, bProcessLastOfLevel, QUAD_SPLIT) + nextSection, TComTU.cpp:88-185
  TU t = *p;
  t.log2 = p->log2 - 1; t.trDepth = p->trDepth + 1; t.parts = p->parts >> 2; if (t.parts < 1) t.parts = 1;
  t.relZ = (int16_t)(p->relZ + section * t.parts); t.section = (int16_t)section;
  t.x = (int16_t)(p->x + (section & 1) * (1 << t.log2)); t.y = (int16_t)(p->y + (section >> 1) * (1 << t.log2));
  if (t.log2 >= 3) {
    t.cW = (int16_t)(1 << (t.log2 - 1)); t.cCodeAll = 1; t.cTrDepth = t.trDepth; t.cRelZ = t.relZ; t.cParts = t.parts;
    t.cOff = (int16_t)((t.cuZ + t.relZ) * 4); t.cx = t.x >> 1; t.cy = t.y >> 1;
  } else { // 4x4 luma: the 4x4 chroma block of the parent is carried by one quadrant
    t.cCodeAll = 0; t.cTrDepth = p->cTrDepth; t.cRelZ = t.relZ & ~3; t.cParts = (int16_t)(t.parts * 4);
    t.cOff = p->cOff; t.cx = p->cx; t.cy = p->cy;
    t.cW = (section == (processLast ? 3 : 0)) ? 4 : 0;
  }
  return t;
}

// ------------------------------------------------------------------------------------------------
// neighbour helpers
// ------------------------------------------------------------------------------------------------
HM_DEV inline const CtuMeta *meta_at(const Shared *e, int x4, int y4, int *z)
{ *z = hm_r2z(((y4 & 15) << 4) | (x4 & 15)); const int ca = (y4 >> 4) * e->wCtu + (x4 >> 4); return ca == e->ctuAddr ? &e->meta : e->fb.meta + ca; }

// TComDataCU::getIntraDirPredictor, TComDataCU.cpp:1513-1586 (luma)
HM_DEV inline int intra_dir_predictor(const Shared *e, int z, int preds[3])
{
  const int r = hm_z2r(z);
  const int x4 = e->ctuX * 16 + (r & 15), y4 = e->ctuY * 16 + (r >> 4);
  int left = DC_IDX, above = DC_IDX, zz;
  if (x4 > 0) { const CtuMeta *m = meta_at(e, x4 - 1, y4, &zz); left = (m->pred[zz] == MODE_INTRA) ? m->dirL[zz] : DC_IDX; }
  if ((y4 & 15) != 0) { const CtuMeta *m = meta_at(e, x4, y4 - 1, &zz); above = (m->pred[zz] == MODE_INTRA) ? m->dirL[zz] : DC_IDX; }
  if (left == above) {
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = PLANAR_IDX; preds[1] = DC_IDX; preds[2] = VER_IDX; }
    return 1;
  }
  preds[0] = left; preds[1] = above;
  if (left && above) preds[2] = PLANAR_IDX; else preds[2] = (left + above) < 2 ? VER_IDX : DC_IDX;
  return 2;
}
// TComDataCU::getCtxSplitFlag, TComDataCU.cpp:1587-1601
HM_DEV inline int ctx_split_flag(const Shared *e, int z, int depth)
{
  const int r = hm_z2r(z);
  const int x4 = e->ctuX * 16 + (r & 15), y4 = e->ctuY * 16 + (r >> 4);
  int ctx = 0, zz;
  if (x4 > 0) { const CtuMeta *m = meta_at(e, x4 - 1, y4, &zz); ctx += m->depth[zz] > depth; }
  if (y4 > 0) { const CtuMeta *m = meta_at(e, x4, y4 - 1, &zz); ctx += m->depth[zz] > depth; }
  return ctx;
}

// ------------------------------------------------------------------------------------------------
// intra reference samples (TComPattern.cpp:107-500)
// ------------------------------------------------------------------------------------------------
HM_DEV inline int avail_above_right(const Shared *e, int rtx4, int rty4, int k)
{ // TComDataCU::getPUAboveRightAdi, TComDataCU.cpp:1302-1360
  if ((rtx4 + k) * 4 >= e->width) return 0;
  const int cx = rtx4 & 15, cy = rty4 & 15;
  if (cx + k <= 15) {
    if (cy != 0) return hm_r2z((cy << 4) | cx) > hm_r2z(((cy - 1) << 4) | (cx + k));
    return rty4 > 0;
  }
  if (cy != 0) return 0;
  return rty4 > 0 && (rtx4 >> 4) < e->wCtu - 1;
}
HM_DEV inline int avail_below_left(const Shared *e, int lbx4, int lby4, int k)
{ // TComDataCU::getPUBelowLeftAdi, TComDataCU.cpp:1244-1300
  if ((lby4 + k) * 4 >= e->height) return 0;
  const int cx = lbx4 & 15, cy = lby4 & 15;
  if (cy + k <= 15) {
    if (cx != 0) return hm_r2z((cy << 4) | cx) > hm_r2z(((cy + k) << 4) | (cx - 1));
    return lbx4 > 0;
  }
  return 0;
}

// TComPrediction::initAdiPatternChType + fillReferenceSamples + smoothing (TComPattern.cpp:107-500).
// (px,py): block position in the component plane; n: block size; (x4,y4): top-left luma 4x4 unit;
// units: block size in 4x4-luma units.  Result in e->u.ref.refTop/refLeft[0] (and [1] when filter != 0).
HM_DEV HM_NOINLINE void init_adi_pattern(Shared *e, int comp, int px, int py, int n, int x4, int y4, int units, int filter)
{
  HM_ENTRY(e); comp = HM_UNI(comp); px = HM_UNI(px); py = HM_UNI(py); n = HM_UNI(n); x4 = HM_UNI(x4); y4 = HM_UNI(y4); units = HM_UNI(units); filter = HM_UNI(filter);
  const int uw = comp ? 2 : 4, uwShift = comp ? 1 : 2, total = 4 * units + 1, L = 2 * units, n2 = 2 * n;
  const int bitDepth = e->bitDepth;
  uint8_t *flags = e->flags;
  // availability flag of every unit: one lane per unit, then a wave reduction for the count
  int cnt = 0;
  HM_PAR_FOR(u, total) {
    int a;
    if (u == L) a = (x4 > 0 && y4 > 0);
    else if (u > L && u <= L + units) a = (y4 > 0);
    else if (u > L + units) a = avail_above_right(e, x4 + units - 1, y4, u - L - units);
    else if (u >= L - units) a = (x4 > 0);
    else a = avail_below_left(e, x4, y4 + units - 1, L - units - u);
    flags[u] = (uint8_t)a; cnt += a;
  }
  const int num = hm_wave_sum_i(cnt);
  HM_SYNC();
  const Pel *rec = e->fb.rec[comp]; const int st = e->stride[comp];
  const int dc = 1 << (bitDepth - 1);
  Pel *top = e->u.ref.refTop[0], *left = e->u.ref.refLeft[0];
  if (num == 0) {
    HM_PAR_FOR(i, n2 + 1) { top[i] = (Pel)dc; left[i] = (Pel)dc; }
  } else if (num == total) {
    HM_PAR_FOR(i, n2 + 1) { top[i] = rec[(py - 1) * st + px - 1 + i]; left[i] = rec[(py - 1 + i) * st + px - 1]; }
  } else {
    // line[]: 2n left samples bottom-to-top, uw copies of the corner, 2n above samples
    Pel *line = e->u.ref.line; const int nl = n2 + uw + n2;
    HM_PAR_FOR(i, nl) {
      int v = dc;
      if (i < n2) { const int j = (n2 - 1 - i); if (flags[L - 1 - (j >> uwShift)]) v = rec[(py + j) * st + px - 1]; }
      else if (i < n2 + uw) { if (flags[L]) v = rec[(py - 1) * st + px - 1]; }
      else { const int j = i - n2 - uw; if (flags[L + 1 + (j >> uwShift)]) v = rec[(py - 1) * st + px + j]; }
      line[i] = (Pel)v;
    }
    HM_SYNC();
    // substitution process (TComPattern.cpp:432-484): serial over at most 65 units, uniform
    int cur = 0;
    if (!flags[0]) {
      int next = 1; while (next < total && !flags[next]) next++;
      const Pel ref = line[next * uw];
      for (; cur < next; cur++) for (int i = 0; i < uw; i++) line[cur * uw + i] = ref;
    }
    for (; cur < total; cur++)
      if (!flags[cur]) { const Pel ref = line[cur * uw - 1]; for (int i = 0; i < uw; i++) line[cur * uw + i] = ref; }
    HM_SYNC();
    HM_PAR_FOR(i, n2 + 1) { top[i] = line[n2 + uw - 1 + i]; left[i] = (i == 0) ? line[n2 + uw - 1] : line[n2 - i]; }
  }
  HM_SYNC();
  if (!filter) return;
  // smoothing, TComPattern.cpp:180-283
  Pel *ft = e->u.ref.refTop[1], *fl = e->u.ref.refLeft[1];
  int strong = (comp == 0);
  const int bl = left[n2], tl = top[0], tr = top[n2];
  if (strong) {
    const int thr = 1 << (bitDepth - 5);
    const int bilLeft = hm_abs((bl + tl) - 2 * left[n]) < thr, bilAbove = hm_abs((tl + tr) - 2 * top[n]) < thr;
    if (n < 32 || !bilLeft || !bilAbove) strong = 0;
  }
  if (strong) {
    const int shift = hm_log2(n) + 1;
    HM_PAR_FOR(i, n2 + 1) {
      if (i == 0) { ft[0] = (Pel)tl; fl[0] = (Pel)tl; }
      else if (i == n2) { ft[n2] = (Pel)tr; fl[n2] = (Pel)bl; }
      else { fl[i] = (Pel)((i * bl + (n2 - i) * tl + n) >> shift); ft[i] = (Pel)(((n2 - i) * tl + i * tr + n) >> shift); }
    }
  } else {
    HM_PAR_FOR(i, n2 + 1) {
      if (i == 0) { const Pel c = (Pel)((left[1] + 2 * top[0] + top[1] + 2) >> 2); ft[0] = c; fl[0] = c; }
      else if (i == n2) { ft[n2] = top[n2]; fl[n2] = left[n2]; }
      else { fl[i] = (Pel)((left[i + 1] + 2 * left[i] + left[i - 1] + 2) >> 2); ft[i] = (Pel)((top[i - 1] + 2 * top[i] + top[i + 1] + 2) >> 2); }
    }
  }
  HM_SYNC();
}

// TComPrediction::filteringIntraReferenceSamples, TComPattern.cpp:514-540
HM_DEV inline int use_filtered_refs(int comp, int mode, int n)
{
  if (comp != 0 || mode == DC_IDX) return 0;
  const int d1 = hm_abs(mode - HOR_IDX), d2 = hm_abs(mode - VER_IDX);
  return (d1 < d2 ? d1 : d2) > HM_INTRA_FILTER[hm_log2(n) - 2];
}

// TComPrediction::predIntraAng (+xPredIntraPlanar/xPredIntraAng/xDCPredFiltering), TComPrediction.cpp:182-840.
// One lane per sample.
HM_DEV HM_NOINLINE void pred_intra(Shared *e, int comp, int mode, int n, int filtered, Pel *dst, int ds)
{
  HM_ENTRY(e); comp = HM_UNI(comp); mode = HM_UNI(mode); n = HM_UNI(n); filtered = HM_UNI(filtered); ds = HM_UNI(ds); dst = hm_uni_ptr(dst);
  const Pel *top = e->u.ref.refTop[filtered], *left = e->u.ref.refLeft[filtered];
  const int bitDepth = e->bitDepth, l2 = hm_log2(n);
  if (mode == PLANAR_IDX) {
    const int bottomLeft = left[n + 1], topRight = top[n + 1];
    HM_PAR_FOR(i, n * n) {
      const int y = i >> l2, x = i & (n - 1);
      // closed form of the running sums of xPredIntraPlanar (integer arithmetic, identical values)
      const int hor = (left[y + 1] << l2) + n + (x + 1) * (topRight - left[y + 1]);
      const int ver = (top[x + 1] << l2) + (y + 1) * (bottomLeft - top[x + 1]);
      dst[y * ds + x] = (Pel)((hor + ver) >> (l2 + 1));
    }
    HM_SYNC();
    return;
  }
  if (mode == DC_IDX) {
    int s = 0;
    HM_PAR_FOR(i, n) s += top[i + 1] + left[i + 1];
    const int dc = (hm_wave_sum_i(s) + n) / (n + n);
    const int edge = (comp == 0 && n <= 16);
    HM_PAR_FOR(i, n * n) {
      const int y = i >> l2, x = i & (n - 1);
      int v = dc;
      if (edge) { // xDCPredFiltering
        if (x == 0 && y == 0) v = (top[1] + left[1] + 2 * dc + 2) >> 2;
        else if (y == 0) v = (top[x + 1] + 3 * dc + 2) >> 2;
        else if (x == 0) v = (left[y + 1] + 3 * dc + 2) >> 2;
      }
      dst[y * ds + x] = (Pel)v;
    }
    HM_SYNC();
    return;
  }
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER_IDX : -(mode - HOR_IDX);
  const int absAng = HM_ANG_TABLE[hm_abs(angMode)], invAngle = HM_INV_ANG_TABLE[hm_abs(angMode)];
  const int angle = angMode < 0 ? -absAng : absAng;
  Pel *refMain = e->u.ref.refMain + 64, *refSide = e->u.ref.refSide + 64;
  if (angle < 0) {
    HM_PAR_FOR(i, n + 1) { refMain[i] = isVer ? top[i] : left[i]; refSide[i] = isVer ? left[i] : top[i]; }
    HM_SYNC();
    const int lastK = (n * angle) >> 5;                 // extend the main reference to the left
    HM_PAR_FOR(j, -1 - lastK) { const int k = -1 - j; refMain[k] = refSide[(128 + (j + 1) * invAngle) >> 8]; }
  } else {
    HM_PAR_FOR(i, 2 * n + 1) { refMain[i] = isVer ? top[i] : left[i]; refSide[i] = isVer ? left[i] : top[i]; }
  }
  HM_SYNC();
  const int edge = (angle == 0 && comp == 0 && n <= 16);
  HM_PAR_FOR(i, n * n) {
    const int y = i >> l2, x = i & (n - 1);   // (x,y) in the prediction's own orientation
    int v;
    if (angle == 0) {
      v = refMain[x + 1];
      if (edge && x == 0) v = hm_clip3(0, (1 << bitDepth) - 1, v + ((refSide[y + 1] - refSide[0]) >> 1));
    } else {
      const int deltaPos = (y + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
      if (df) v = ((32 - df) * refMain[x + di + 1] + df * refMain[x + di + 2] + 16) >> 5;
      else v = refMain[x + di + 1];
    }
    if (isVer) dst[y * ds + x] = (Pel)v; else dst[x * ds + y] = (Pel)v;
  }
  HM_SYNC();
}

// One predicted sample of `mode` at (x,y), straight from the reference lines in LDS: the same arithmetic as
// pred_intra above, evaluated per sample so that a lane can own a whole (mode, 8x8 block) SATD task.
HM_DEV inline int pred_sample(const Shared *e, int mode, int n, int l2, int x, int y, int dcVal, int bitDepth)
{
  const int filt = use_filtered_refs(0, mode, n);
  const Pel *top = e->u.ref.refTop[filt], *left = e->u.ref.refLeft[filt];
  if (mode == PLANAR_IDX) {
    const int hor = (left[y + 1] << l2) + n + (x + 1) * (top[n + 1] - left[y + 1]);
    const int ver = (top[x + 1] << l2) + (y + 1) * (left[n + 1] - top[x + 1]);
    return (hor + ver) >> (l2 + 1);
  }
  if (mode == DC_IDX) {
    if (n <= 16) {
      if (x == 0 && y == 0) return (top[1] + left[1] + 2 * dcVal + 2) >> 2;
      if (y == 0) return (top[x + 1] + 3 * dcVal + 2) >> 2;
      if (x == 0) return (left[y + 1] + 3 * dcVal + 2) >> 2;
    }
    return dcVal;
  }
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER_IDX : -(mode - HOR_IDX);
  const int absAng = HM_ANG_TABLE[hm_abs(angMode)], invAngle = HM_INV_ANG_TABLE[hm_abs(angMode)];
  const int angle = angMode < 0 ? -absAng : absAng;
  const Pel *mainR = isVer ? top : left, *sideR = isVer ? left : top;
  const int xx = isVer ? x : y, yy = isVer ? y : x;
  if (angle == 0) {
    int v = mainR[xx + 1];
    if (n <= 16 && xx == 0) v = hm_clip3(0, (1 << bitDepth) - 1, v + ((sideR[yy + 1] - sideR[0]) >> 1));
    return v;
  }
  const int deltaPos = (yy + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
  const int i0 = xx + di + 1;
  const int a = i0 >= 0 ? mainR[i0] : sideR[(128 - i0 * invAngle) >> 8];
  if (!df) return a;
  const int i1 = i0 + 1;
  const int b = i1 >= 0 ? mainR[i1] : sideR[(128 - i1 * invAngle) >> 8];
  return ((32 - df) * a + df * b + 16) >> 5;
}

// SATD of all 35 modes of an n x n luma PU (n <= 16): lanes own (mode, 8x8 block) tasks and predict their
// samples on the fly, so the whole first-pass mode estimation of a small PU is one or a few wave passes.
HM_DEV HM_NOINLINE void satd_all_modes_small(Shared *e, const Pel *org, int so, int n)
{
  HM_ENTRY(e); so = HM_UNI(so); n = HM_UNI(n); org = hm_uni_ptr(org);
  const int l2 = hm_log2(n), bitDepth = e->bitDepth;
  int s = 0;
  HM_PAR_FOR(i, n) s += e->u.ref.refTop[0][i + 1] + e->u.ref.refLeft[0][i + 1];
  const int dcVal = (hm_wave_sum_i(s) + n) / (n + n);
  HM_PAR_FOR(i, 36) e->satd[i] = 0;
  HM_SYNC();
  if (n == 4) {
    HM_PAR_FOR(mode, 35) {
      int d[16];
#pragma unroll
      for (int i = 0; i < 16; i++) d[i] = org[(i >> 2) * so + (i & 3)] - pred_sample(e, mode, 4, 2, i & 3, i >> 2, dcVal, bitDepth);
#pragma unroll
      for (int y = 0; y < 4; y++) {
        const int a = d[y * 4] + d[y * 4 + 1], b = d[y * 4] - d[y * 4 + 1], c = d[y * 4 + 2] + d[y * 4 + 3], f = d[y * 4 + 2] - d[y * 4 + 3];
        d[y * 4] = a + c; d[y * 4 + 1] = b + f; d[y * 4 + 2] = a - c; d[y * 4 + 3] = b - f;
      }
      uint32_t t = 0;
#pragma unroll
      for (int x = 0; x < 4; x++) {
        const int a = d[x] + d[4 + x], b = d[x] - d[4 + x], c = d[8 + x] + d[12 + x], f = d[8 + x] - d[12 + x];
        t += (uint32_t)(hm_abs(a + c) + hm_abs(b + f) + hm_abs(a - c) + hm_abs(b - f));
      }
      e->satd[mode] = ((t + 1) >> 1) >> (bitDepth - 8);
    }
  } else {
    const int nb = n >> 3;
    HM_PAR_FOR(task, 35 * nb * nb) {
      const int mode = task >> (2 * (l2 - 3)), b = task & (nb * nb - 1), by = (b >> (l2 - 3)) * 8, bx = (b & (nb - 1)) * 8;
      int d[64];
#pragma unroll
      for (int i = 0; i < 64; i++) d[i] = org[(by + (i >> 3)) * so + bx + (i & 7)] - pred_sample(e, mode, n, l2, bx + (i & 7), by + (i >> 3), dcVal, bitDepth);
#pragma unroll
      for (int y = 0; y < 8; y++) {
#pragma unroll
        for (int len = 1; len < 8; len <<= 1)
#pragma unroll
          for (int bb = 0; bb < 8; bb += len << 1)
#pragma unroll
            for (int k = 0; k < len; k++) { const int a0 = d[y * 8 + bb + k], a1 = d[y * 8 + bb + k + len]; d[y * 8 + bb + k] = a0 + a1; d[y * 8 + bb + k + len] = a0 - a1; }
      }
      uint32_t t = 0;
#pragma unroll
      for (int x = 0; x < 8; x++) {
#pragma unroll
        for (int len = 1; len < 8; len <<= 1)
#pragma unroll
          for (int bb = 0; bb < 8; bb += len << 1)
#pragma unroll
            for (int k = 0; k < len; k++) { const int a0 = d[(bb + k) * 8 + x], a1 = d[(bb + k + len) * 8 + x]; d[(bb + k) * 8 + x] = a0 + a1; d[(bb + k + len) * 8 + x] = a0 - a1; }
#pragma unroll
        for (int y = 0; y < 8; y++) t += (uint32_t)hm_abs(d[y * 8 + x]);
      }
      HM_LDS_ADD(&e->satd[mode], (t + 2) >> 2);
    }
    HM_SYNC();
    HM_PAR_FOR(mode, 35) e->satd[mode] >>= (bitDepth - 8);
  }
  HM_SYNC();
}

// ------------------------------------------------------------------------------------------------
// coefficient coding parameters
// ------------------------------------------------------------------------------------------------
HM_DEV inline int coef_scan_idx(const CtuMeta *m, int z, int n, int comp)
{
  if (m->pred[z] != MODE_INTRA) return SCAN_DIAG; // TComDataCU::getCoefScanIdx, TComDataCU.cpp:3340-3380
  if (n > (comp ? 4 : 8)) return SCAN_DIAG;
  int dir = comp ? m->dirC[z] : m->dirL[z];
  if (dir == DM_CHROMA_IDX) dir = m->dirL[z & ~3];
  if (hm_abs(dir - VER_IDX) <= 4) return SCAN_HOR;
  if (hm_abs(dir - HOR_IDX) <= 4) return SCAN_VER;
  return SCAN_DIAG;
}
HM_DEV inline int first_sig_ctx(int n, int scanType, int chroma)
{ // getTUEntropyCodingParameters, TComChromaFormat.cpp:75-130
  if (n == 4) return 0;
  if (n == 8) return 9 + ((scanType != SCAN_DIAG && !chroma) ? 6 : 0);
  return chroma ? 12 : 21;
}
HM_DEV inline int sig_ctx_inc(int pattern, int firstCtx, int blkPos, int log2n, int chroma)
{ // TComTrQuant::getSigCtxInc, TComTrQuant.cpp:2548-2640
  const int posY = blkPos >> log2n, posX = blkPos - (posY << log2n);
  if (posX + posY == 0) return 0;
  int offset;
  if (log2n == 2) offset = (int)((0x8877886654325410ull >> (4 * (4 * posY + posX))) & 15);      // ctxIndMap4x4 (HM_CTX_IND_MAP_4x4), 16 nibbles: no table read per lane
  else {
    int cnt; const int xs = posX & 3, ys = posY & 3;
    if (pattern == 0) cnt = (xs + ys >= 3) ? 0 : ((xs + ys >= 1) ? 1 : 2);
    else if (pattern == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
    else if (pattern == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
    else cnt = 2;
    const int notFirst = ((posX >> 2) + (posY >> 2)) > 0;
    offset = ((notFirst && !chroma) ? 3 : 0) + cnt;
  }
  return firstCtx + offset;
}
HM_DEV inline int pattern_sig_ctx(const uint8_t *cgFlag, int cgx, int cgy, int wg)
{ // TComTrQuant::calcPatternSigCtx, TComTrQuant.cpp:2522-2535
  if (wg <= 1) return 0;
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgFlag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgFlag[(cgy + 1) * wg + cgx] != 0;
  return r + (l << 1);
}
HM_DEV inline int sig_cg_ctx(const uint8_t *cgFlag, int cgx, int cgy, int wg)
{ // TComTrQuant::getSigCoeffGroupCtxInc, TComTrQuant.cpp:2872-2886
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgFlag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgFlag[(cgy + 1) * wg + cgx] != 0;
  return (r + l) != 0;
}
HM_DEV inline int ctx_set_index(int chroma, int subset, int gt1)
{ return (chroma ? 4 : 0) + ((!chroma && subset > 0) ? 2 : 0) + (gt1 ? 1 : 0); }   // TComChromaFormat.h:243
HM_DEV inline void last_ctx_params(int chroma, int n, int *off, int *shift)
{ // getLastSignificantContextParameters, TComChromaFormat.h:211
  const int cv = hm_log2(n) - 2;
  *off = chroma ? 0 : (cv * 3 + ((cv + 1) >> 2));
  *shift = chroma ? cv : ((cv + 3) >> 2);
}

// ------------------------------------------------------------------------------------------------
// RDOQ (TComTrQuant::xRateDistOptQuant, TComTrQuant.cpp:1974-2511).  Source coefficients in LDS bufA
// (stride HM_TSTRIDE); levels go to dst (dense n*n, HBM).  The level decision is a serial chain over
// the scan (running c1/c2/Rice state), evaluated wave-uniformly.
// ------------------------------------------------------------------------------------------------
HM_DEV inline int hm_group_idx(int v)
{ // g_uiGroupIdx (TComRom.cpp:318) in closed form
  if (v < 4) return v;
  const int msb = 31 - __builtin_clz((unsigned)v);
  return 2 * msb + ((v >> (msb - 1)) & 1);
}
// bit-cost tables of one RDOQ call: one entry per lane (lane variables), index = (context - group base) * 2 + bin
//   tSig  the 28 (luma) / 16 (chroma) significance contexts of the component
//   tOne  the 24 greater-than-1 contexts, then the 6 greater-than-2 contexts (lanes 48..59)
//   tLast the 15 last-X contexts of the component, then its 15 last-Y contexts (lanes 30..59)
//   tMisc the 4 coded-sub-block contexts, then the 10 cbf contexts (lanes 8..27)
HM_DEV inline int ic_rate(HM_LVARG(int32_t, tOne), uint32_t absLevel, int ctxOne, int ctxAbs, int goRice, int c1Idx, int c2Idx)
{ // xGetICRate, TComTrQuant.cpp:2725-2800
  int rate = 32768;
  const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < (3u << goRice)) { length = symbol >> goRice; rate += (int)((length + 1 + goRice) << 15); }
    else {
      length = (uint32_t)goRice; symbol -= (3u << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((3 + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < 8) { rate += HM_LV_GET(tOne, ctxOne * 2 + 1); if (c2Idx < 1) rate += HM_LV_GET(tOne, 48 + ctxAbs * 2 + 1); }
  } else if (absLevel == 1) rate += HM_LV_GET(tOne, ctxOne * 2);
  else if (absLevel == 2) { rate += HM_LV_GET(tOne, ctxOne * 2 + 1); rate += HM_LV_GET(tOne, 48 + ctxAbs * 2); }
  else rate = 0;
  return rate;
}

// Structure on the wavefront:
//   1. lane-parallel pre-pass over the scan: |c|*scale, sign, raster position -> LDS; wave-max gives the
//      last significant scan position, all-zero blocks leave here;
//   2. the bit costs of the contexts RDOQ can touch are tabulated for the current estimator state in four
//      lane variables (one VGPR each) and read back with v_readlane / ds_bpermute: the serial parts below
//      touch no memory;
//   3. per coefficient group the 16 positions are loaded and costed lane-parallel under the zero-level
//      hypothesis; the level decision is the reference's serial chain (running c1/c2/Rice state, double
//      sums in the reference's operation order) evaluated wave-uniformly on lane-variable data; only the
//      positions with a non-zero quantised magnitude run the level search;
//   4. the last-position search and sign-bit hiding work the same way: lane-parallel per-group precompute,
//      then a wave-uniform serial pass in the reference's order.
// Nothing to decide when even the largest coefficient quantises to level 0: rdoq's pre-pass would find no position with a non-zero rounded level
// and return 0.  Checked at the call site, in one pass over the block in LDS -- a call of the big stage (register saves and restores through
// private memory) costs more than the check, and 98 % of the blocks of an inter residual quadtree are empty.  The levels are not written
// then: every caller clears them itself when a block comes back empty.
HM_DEV inline int rdoq_is_empty(const Shared *e, int n, int comp)
{
  const int chroma = comp != 0, log2n = hm_log2(n);
  const int qBits = 14 + e->fb.qpPer[chroma] + (15 - e->bitDepth - log2n);
  const int quantCoef = HM_QUANT_SCALES[e->fb.qpRem[chroma]];
  int mx = 0;
  HM_PAR_FOR(i, n * n) { const int a = hm_abs(e->bufA[(i >> log2n) * HM_TSTRIDE + (i & (n - 1))]); mx = a > mx ? a : mx; }
  mx = hm_wave_max_i(mx);
  const int64_t cap = 2147483647LL - (1LL << (qBits - 1)), t = (int64_t)mx * quantCoef;
  const int32_t lvl = (int32_t)(t < cap ? t : cap);
  return ((lvl + (1 << (qBits - 1))) >> qBits) == 0;
}
HM_DEV HM_NOINLINE int rdoq(Shared *e, TCoeff *dst, int n, int comp, int scanType, int cbfCtx)
{
  HM_ENTRY(e); n = HM_UNI(n); comp = HM_UNI(comp); scanType = HM_UNI(scanType); cbfCtx = HM_UNI(cbfCtx); dst = hm_uni_ptr(dst);
  const Cabac *cb = &e->cur;
  const int chroma = comp != 0, log2n = hm_log2(n), bitDepth = e->bitDepth;
  const double lambda = chroma ? e->fb.lambdaC : e->fb.lambda;
  const int transformShift = 15 - bitDepth - log2n;
  const int qBits = 14 + e->fb.qpPer[chroma] + transformShift;
  const int quantCoef = HM_QUANT_SCALES[e->fb.qpRem[chroma]];
  const double errScale = e->fb.errScale[chroma][log2n - 2];
  const int numCoef = n * n, wg = n >> 2;
  const uint16_t *scan = e->tab->scan[scanType][log2n - 2], *scanCG = e->tab->scanCG[scanType][log2n - 2];
  const int firstCtx = first_sig_ctx(n, scanType, chroma);
  const int sigOff = C_SIG + (chroma ? 28 : 0);
  const int32_t *src = e->bufA;
  WorkSpace *ws = e->ws;
  // Per-position state, indexed by scan position.  Blocks up to 16x16: the RqLds arrays + rows 16..31 of bufA for the costs.  32x32 blocks
  // (`big`) keep it in LDS too, in 8 bytes per position: |c|*scale in the transform temp (u.bufB), raster position | sign | significance code
  // and the decided level in bufA once the pre-pass has read the coefficients out of it; the working level (decision, or 0 in a zeroed
  // group, cut at the chosen last position, signed) and the significance context are derived where they are needed, and only the cost of
  // the non-zero levels goes through the HBM workspace (fetched one group ahead).
  const bool big = n == 32;
  double *costCoeff = big ? ws->costCoeff : (double *)(e->bufA + 16 * HM_TSTRIDE);
  int32_t *rqLvl = big ? (int32_t *)e->u.bufB : e->u.rq.lvl;
  uint16_t *rqPos = big ? (uint16_t *)e->bufA : e->u.rq.pos, *rqDec = big ? (uint16_t *)e->bufA + 1024 : e->u.rq.dec;
  int16_t *rqCur = e->u.rq.cur;                              // not used by 32x32 blocks
  uint8_t *rqCtxSig = e->u.rq.ctxSig, *rqCode = e->u.rq.code;   // not used by 32x32 blocks
  uint8_t *cgCtxSet = big ? (uint8_t *)e->u.bufB + 4096 : e->u.rq.cgCtxSet;
  // ---- 1. pre-pass
  int lastLocal = -1;
  {
    const int64_t cap = 2147483647LL - (1LL << (qBits - 1));
    HM_PAR_FOR(sp, numCoef) {
      const int blkPos = scan[sp];
      const int32_t sc = src[(blkPos >> log2n) * HM_TSTRIDE + (blkPos & (n - 1))];
      const int64_t tmpLevel = (int64_t)hm_abs(sc) * quantCoef;
      const int32_t lvl = (int32_t)(tmpLevel < cap ? tmpLevel : cap);
      rqLvl[sp] = (big && sc < 0) ? ~lvl : lvl;               // 32x32: the sign rides along until the positions can go where the coefficients are
      if (!big) rqPos[sp] = (uint16_t)(blkPos | (sc < 0 ? 0x8000 : 0));
      if (((lvl + (1 << (qBits - 1))) >> qBits) > 0 && sp > lastLocal) lastLocal = sp;
      dst[sp] = 0;
    }
  }
  const int lastScanPos = hm_wave_max_i(lastLocal);
  HM_SYNC();
  if (big && lastScanPos >= 0) {
    HM_PAR_FOR(sp, numCoef) {
      const int32_t v = rqLvl[sp]; const int neg = v < 0;
      rqLvl[sp] = neg ? ~v : v;
      rqPos[sp] = (uint16_t)(scan[sp] | (neg ? 0x8000 : 0));
      rqDec[sp] = 0;
    }
    HM_SYNC();
  }
  if (lastScanPos < 0) return 0;
  // ---- 2. bit-cost tables of the current estimator state (TEncSbac::estBit, TEncSbac.cpp:1717-1956)
  int lastOff, lastShift; last_ctx_params(chroma, n, &lastOff, &lastShift);
  HM_LV(int32_t, tSig); HM_LV(int32_t, tOne); HM_LV(int32_t, tLast); HM_LV(int32_t, tMisc); HM_LV(int32_t, tLastCost);
  HM_LV(double, vCGSig);                  // cost of the coded-sub-block flag per coefficient group (scan order)
  HM_WAVE_FOR(k) {
    const int bin = k & 1, c = k >> 1;
    HM_LVK(tSig, k) = c < (chroma ? 16 : 28) ? HM_LT()->ebits[cb->s[sigOff + c] ^ bin] : 0;
    HM_LVK(tOne, k) = c < 30 ? HM_LT()->ebits[cb->s[C_ONE + c] ^ bin] : 0;                   // C_ABS follows C_ONE
    HM_LVK(tLast, k) = c < 30 ? HM_LT()->ebits[cb->s[(c < 15 ? C_LASTX + c : C_LASTY + c - 15) + (chroma ? 15 : 0)] ^ bin] : 0;
    HM_LVK(tMisc, k) = c < 15 ? HM_LT()->ebits[cb->s[c < 4 ? C_SIG_CG + c : (c < 14 ? C_QT_CBF + c - 4 : C_ROOT_CBF)] ^ bin] : 0;   // cbfCtx 10 = root cbf
    HM_LVK(tLastCost, k) = 0;
    HM_LVK(vCGSig, k) = 0;
  }
  { // bits of a last-position group index (xGetRateLast, TComTrQuant.cpp:2815-2832 over estLastSignificantPositionBit,
    // TEncSbac.cpp:1846-1892): prefix + bypass bits; lane g = X component, lane 16+g = Y component
    const int gmax = HM_GROUP_IDX[n - 1];
    int accX = 0, accY = 0;
    for (int g = 0; g <= gmax; g++) {
      const int cx = (lastOff + (g >> lastShift)) * 2, cy = 30 + cx;
      int bx = accX, by = accY;
      if (g < gmax) { bx += HM_LV_GET(tLast, cx); by += HM_LV_GET(tLast, cy); }
      if (g > 3) { bx += 32768 * ((g - 2) >> 1); by += 32768 * ((g - 2) >> 1); }
      HM_LV_SET(tLastCost, g, bx); HM_LV_SET(tLastCost, 16 + g, by);
      accX += HM_LV_GET(tLast, cx + 1); accY += HM_LV_GET(tLast, cy + 1);
    }
  }
  // ---- 3. decision chain
  double blockUncodedCost = 0;
  // distortion of the positions behind the last significant one: only non-zero terms change the sum
  for (int base = numCoef > 64 ? numCoef - 64 : 0; base >= 0 && base + 63 > lastScanPos; base -= 64) {
    HM_LV(double, vT); uint64_t m = 0;
    HM_WAVE_FOR(k) {
      const int sp = base + k;
      const int32_t lvl = (sp < numCoef && sp > lastScanPos) ? rqLvl[sp] : 0;
      const double err = (double)lvl;
      HM_LVK(vT, k) = err * err * errScale;
      HM_BALLOT(m, k, lvl != 0);
    }
    while (m) { const int k = 63 - __builtin_clzll(m); blockUncodedCost += HM_LV_GETD(vT, k); m &= ~(1ull << k); }
  }
  double baseCost = blockUncodedCost;
  uint64_t cgMask = 0;                    // significant-coefficient-group flags, bit = raster position of the group
  const int cgLastScanPos = lastScanPos >> 4;
  int ctxSet = ctx_set_index(chroma, lastScanPos >> 4, 0), c1 = 1, c2 = 0, c1Idx = 0, c2Idx = 0, goRice = 0;
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0; cgScanPos--) {
    const int cgBlkPos = scanCG[cgScanPos], cgy = cgBlkPos >> (log2n - 2), cgx = cgBlkPos & (wg - 1);
    const uint64_t cgBit = 1ull << cgBlkPos;
    double sigCost = 0, sigCost0 = 0, codedLevelAndDist = 0, uncodedDist = 0; int nnzBeforePos0 = 0;
    const int sigRight = (cgx < wg - 1) ? (int)((cgMask >> (cgBlkPos + 1)) & 1) : 0;       // calcPatternSigCtx, TComTrQuant.cpp:2522
    const int sigLower = (cgy < wg - 1) ? (int)((cgMask >> (cgBlkPos + wg)) & 1) : 0;
    const int pattern = wg <= 1 ? 0 : sigRight + (sigLower << 1);
    const int startPos = (cgScanPos == cgLastScanPos ? (lastScanPos & 15) : 15);
    const int wSet = ctxSet;              // context set this group starts with (sign-bit hiding walks it again)
    HM_LV(int32_t, vLvl); HM_LV(int32_t, vMx); HM_LV(int32_t, vSigIdx); HM_LV(int32_t, vB0); HM_LV(int32_t, vB1);
    HM_LV(int32_t, vDec); HM_LV(int32_t, vCode);
    HM_LV(double, vC0); HM_LV(double, vS0); HM_LV(double, vCoef0); HM_LV(double, vCC);
    uint64_t nzMask = 0;
    HM_WAVE_FOR(k) {
      const int scanPos = cgScanPos * 16 + (k & 15), blkPos = rqPos[scanPos] & 0x3ff;
      const int inRange = (k & 15) <= startPos;                   // positions behind the last significant one add +0.0
      const int32_t lvl = rqLvl[scanPos];
      uint32_t mx = (uint32_t)((lvl + (1 << (qBits - 1))) >> qBits); if (mx > 32767u) mx = 32767u;
      const double err = (double)lvl, c0 = inRange ? err * err * errScale : 0.0;
      const int sigIdx = (scanPos == lastScanPos) ? 0 : sig_ctx_inc(pattern, firstCtx, blkPos, log2n, chroma);
      const int b0 = HM_LV_GATHER(tSig, sigIdx * 2), b1 = HM_LV_GATHER(tSig, sigIdx * 2 + 1);
      const double s0 = inRange ? lambda * (double)b0 : 0.0;
      HM_LVK(vLvl, k) = lvl; HM_LVK(vMx, k) = (int32_t)mx; HM_LVK(vSigIdx, k) = sigIdx; HM_LVK(vB0, k) = b0; HM_LVK(vB1, k) = b1;
      HM_LVK(vC0, k) = c0; HM_LVK(vS0, k) = s0; HM_LVK(vCoef0, k) = c0 + s0; HM_LVK(vCC, k) = 0;
      HM_LVK(vDec, k) = 0; HM_LVK(vCode, k) = 1;
      HM_BALLOT(nzMask, k, mx > 0 && k < 16);
    }
    // level decisions: only the positions with a non-zero quantised magnitude take part in the context chain; their
    // final cost replaces the zero-hypothesis cost in the lane variables (vCoef0 / vS0)
    for (uint64_t todo = nzMask; todo;) {
      const int posInCG = 63 - __builtin_clzll(todo); todo &= ~(1ull << posInCG);
      const int scanPos = cgScanPos * 16 + posInCG;
      uint32_t level = 0;
      double cCoeff, cSig;
      {
        const uint32_t maxAbsLevel = (uint32_t)HM_LV_GET(vMx, posInCG);
        const int32_t levelDouble = HM_LV_GET(vLvl, posInCG);
        const int ctxOne = 4 * ctxSet + c1, ctxAbs = ctxSet;
        const int isLast = (scanPos == lastScanPos);
        int sigBits = 0, sigCode = 0;
        { // xGetCodedLevel, TComTrQuant.cpp:2660-2715
          int currSigBits = 0;
          if (!isLast && maxAbsLevel < 3) { sigBits = HM_LV_GET(vB0, posInCG); sigCode = 1; cCoeff = HM_LV_GETD(vCoef0, posInCG); }
          else cCoeff = HM_MAX_DOUBLE;
          double currCostSig = 0;
          if (!isLast) { currSigBits = HM_LV_GET(vB1, posInCG); currCostSig = lambda * (double)currSigBits; }
          const uint32_t minAbs = maxAbsLevel > 1 ? maxAbsLevel - 1 : 1;
          for (int al = (int)maxAbsLevel; al >= (int)minAbs; al--) {
            const double de = (double)(levelDouble - (int32_t)((uint32_t)al << qBits));
            const double dist = de * de * errScale;
            const double rc = lambda * (double)ic_rate(tOne, (uint32_t)al, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
            double cc = dist + rc;
            cc += currCostSig;
            if (cc < cCoeff) { level = (uint32_t)al; cCoeff = cc; sigBits = currSigBits; sigCode = isLast ? 0 : 2; }
          }
        }
        cSig = lambda * (double)sigBits;
        HM_LV_SET(vCode, posInCG, sigCode); HM_LV_SET(vDec, posInCG, (int32_t)level);
        HM_LV_SETD(vCoef0, posInCG, cCoeff); HM_LV_SETD(vS0, posInCG, cSig);
        const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel && level > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
      }
      if (level) {
        HM_LV_SETD(vCC, posInCG, cCoeff);
        cgMask |= cgBit;
        codedLevelAndDist += cCoeff - cSig;
        uncodedDist += HM_LV_GETD(vC0, posInCG);
        if (posInCG != 0) nnzBeforePos0++;
      }
    }
    if (cgScanPos > 0) {                   // the next group starts a fresh context set
      ctxSet = ctx_set_index(chroma, (cgScanPos * 16 - 1) >> 4, c1 == 0);
      c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0;
    }
    // the three running sums of the reference, position by position from 15 down to 0
    HM_ORDERED_ADD16(baseCost, vCoef0);
    HM_ORDERED_ADD16(blockUncodedCost, vC0);
    if (cgScanPos) { HM_ORDERED_ADD16(sigCost, vS0); sigCost0 = HM_LV_GETD(vS0, 0); }
    int zeroed = 0;
    if (cgScanPos) {
      const int cgCtx = ((sigRight + sigLower) != 0) * 2;                                   // getSigCoeffGroupCtxInc, TComTrQuant.cpp:2872
      if (!(cgMask & cgBit)) {
        const double r0 = lambda * (double)HM_LV_GET(tMisc, (chroma ? 4 : 0) + cgCtx);
        baseCost += r0 - sigCost;
        HM_LV_SETD(vCGSig, cgScanPos, r0);
      } else if (cgScanPos < cgLastScanPos) {
        if (nnzBeforePos0 == 0) { baseCost -= sigCost0; sigCost -= sigCost0; }
        double costZeroCG = baseCost;
        const double r0 = lambda * (double)HM_LV_GET(tMisc, (chroma ? 4 : 0) + cgCtx), r1 = lambda * (double)HM_LV_GET(tMisc, (chroma ? 4 : 0) + cgCtx + 1);
        baseCost += r1;
        costZeroCG += r0;
        costZeroCG += uncodedDist; costZeroCG -= codedLevelAndDist; costZeroCG -= sigCost;
        if (costZeroCG < baseCost) { cgMask &= ~cgBit; baseCost = costZeroCG; HM_LV_SETD(vCGSig, cgScanPos, r0); zeroed = 1; }
        else HM_LV_SETD(vCGSig, cgScanPos, r1);
      }
    } else cgMask |= cgBit;
    // park the group's decisions for the passes below
    HM_WAVE_FOR(k) {
      if (k < 16) {
        const int scanPos = cgScanPos * 16 + k, dec = HM_LVK(vDec, k);
        const int code = (zeroed && dec) ? 0 : HM_LVK(vCode, k);
        rqDec[scanPos] = (uint16_t)dec;
        if (big) rqPos[scanPos] = (uint16_t)(rqPos[scanPos] | (code << 12));
        else { rqCur[scanPos] = (int16_t)(zeroed ? 0 : dec); rqCode[scanPos] = (uint8_t)code; rqCtxSig[scanPos] = (uint8_t)HM_LVK(vSigIdx, k); }
        costCoeff[scanPos] = HM_LVK(vCC, k);
      }
    }
    cgCtxSet[cgScanPos] = (uint8_t)wSet;
  }
  HM_SYNC();
  double bestCost = blockUncodedCost + lambda * (double)HM_LV_GET(tMisc, 8 + cbfCtx * 2);   // TComTrQuant.cpp:2310-2316
  baseCost += lambda * (double)HM_LV_GET(tMisc, 8 + cbfCtx * 2 + 1);
  int bestLastIdxP1 = 0, foundLast = 0;
  HM_LV(double, ccNext);                 // the group's non-zero level costs, fetched one group ahead (HBM for 32x32 blocks)
  HM_WAVE_FOR(k) { HM_LVK(ccNext, k) = costCoeff[cgLastScanPos * 16 + (k & 15)]; }
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0 && !foundLast; cgScanPos--) {
    const int cgBlkPos = scanCG[cgScanPos];
    HM_LV(double, ccCur);
    HM_WAVE_FOR(k) { HM_LVK(ccCur, k) = HM_LVK(ccNext, k); if (cgScanPos > 0) HM_LVK(ccNext, k) = costCoeff[(cgScanPos - 1) * 16 + (k & 15)]; }
    baseCost -= HM_LV_GETD(vCGSig, cgScanPos);
    if (!((cgMask >> cgBlkPos) & 1)) continue;
    const int cgyB = cgBlkPos >> (log2n - 2), cgxB = cgBlkPos & (wg - 1);
    const int patternB = wg <= 1 ? 0 : ((cgxB < wg - 1) ? (int)((cgMask >> (cgBlkPos + 1)) & 1) : 0) + (((cgyB < wg - 1) ? (int)((cgMask >> (cgBlkPos + wg)) & 1) : 0) << 1);
    // per position: baseCost = (baseCost - A) + B with A = cost of the coded level (or of the significance flag of a
    // zero), B = distortion of the uncoded level (or +0.0); a coded position first offers itself as the last one
    HM_LV(double, dSig); HM_LV(double, dLast); HM_LV(double, dA); HM_LV(double, dB); HM_LV(double, dPre);
    uint64_t curMask = 0, gt1Mask = 0;
    HM_WAVE_FOR(k) {
      const int scanPos = cgScanPos * 16 + (k & 15);
      const int inRange = scanPos <= lastScanPos;
      const int posv = rqPos[scanPos], blkPos = posv & 0x3ff;
      // 32x32: a group that gets here kept its decisions; its significance contexts follow from the final group flags (the right and lower
      // neighbours were settled before this group's decisions were taken)
      const int cur = big ? (int)rqDec[scanPos] : (int)rqCur[scanPos], code = big ? ((posv >> 12) & 3) : (int)rqCode[scanPos];
      const int sigIdx = big ? ((scanPos == lastScanPos) ? 0 : sig_ctx_inc(patternB, firstCtx, blkPos, log2n, chroma)) : (int)rqCtxSig[scanPos];
      const int sb = HM_LV_GATHER(tSig, sigIdx * 2 + (code ? code - 1 : 0));
      const double cSig = lambda * (double)(code ? sb : 0);
      HM_LVK(dSig, k) = cSig;
      int posY = blkPos >> log2n, posX = blkPos - (posY << log2n);
      if (scanType == SCAN_VER) { const int t = posX; posX = posY; posY = t; }
      const int lb = HM_LV_GATHER(tLastCost, hm_group_idx(posX)) + HM_LV_GATHER(tLastCost, 16 + hm_group_idx(posY));
      HM_LVK(dLast, k) = lambda * (double)lb;
      const double err = (double)rqLvl[scanPos];
      HM_LVK(dA, k) = !inRange ? 0.0 : (cur ? HM_LVK(ccCur, k) : cSig);
      HM_LVK(dB, k) = (inRange && cur) ? err * err * errScale : 0.0;
      HM_LVK(dPre, k) = 0;
      HM_BALLOT(curMask, k, cur != 0 && k < 16);
      HM_BALLOT(gt1Mask, k, cur > 1 && k < 16);
    }
    HM_ORDERED_CHAIN16(baseCost, dA, dB, dPre);                  // dPre[k] = baseCost before position k
    for (uint64_t todo = curMask; todo;) {
      const int posInCG = 63 - __builtin_clzll(todo); todo &= ~(1ull << posInCG);
      const double t1 = HM_LV_GETD(dPre, posInCG) + HM_LV_GETD(dLast, posInCG);
      const double totalCost = t1 - HM_LV_GETD(dSig, posInCG);
      if (totalCost < bestCost) { bestLastIdxP1 = cgScanPos * 16 + posInCG + 1; bestCost = totalCost; }
      if ((gt1Mask >> posInCG) & 1) { foundLast = 1; break; }
    }
  }
  // ---- levels with signs, truncated at the chosen last position (lane-parallel)
  int absPart = 0;
  HM_PAR_FOR(sp, lastScanPos + 1) {
    int lv = 0;
    if (sp < bestLastIdxP1) lv = big ? (((cgMask >> scanCG[sp >> 4]) & 1) ? (int)rqDec[sp] : 0) : (int)rqCur[sp];
    absPart += lv;
    if (!big) rqCur[sp] = (int16_t)((rqPos[sp] & 0x8000) ? -lv : lv);
  }
  const int absSum = hm_wave_sum_i(absPart);
  HM_SYNC();
  // ---- 4. sign bit hiding, TComTrQuant.cpp:2380-2510
  if (absSum >= 2) {
    const int64_t rdFactor = e->fb.rdFactor[chroma];
    int lastCG = -1;
    for (int subSet = cgLastScanPos; subSet >= 0; subSet--) {
      const int subPos = subSet << 4;
      const int top = (subPos + 15 <= lastScanPos) ? 15 : (lastScanPos - subPos);
      HM_LV(int32_t, sCur); HM_LV(int32_t, sDec); HM_LV(int32_t, sLvl); HM_LV(int32_t, sNeg); HM_LV(int32_t, sDelta);
      uint64_t nzM = 0, oddM = 0;
      const int cgBlkPosS = scanCG[subSet], cgKept = (int)((cgMask >> cgBlkPosS) & 1);
      const int cgyS = cgBlkPosS >> (log2n - 2), cgxS = cgBlkPosS & (wg - 1);
      const int patternS = wg <= 1 ? 0 : ((cgxS < wg - 1) ? (int)((cgMask >> (cgBlkPosS + 1)) & 1) : 0) + (((cgyS < wg - 1) ? (int)((cgMask >> (cgBlkPosS + wg)) & 1) : 0) << 1);
      HM_WAVE_FOR(k) {
        const int kk = k & 15, sp = subPos + kk;
        const int posv = rqPos[sp], dec = rqDec[sp];
        int cur, sigIdx;
        if (big) {
          const int lv = (cgKept && sp < bestLastIdxP1) ? dec : 0;
          cur = (kk <= top) ? ((posv & 0x8000) ? -lv : lv) : 0;
          sigIdx = (sp == lastScanPos) ? 0 : sig_ctx_inc(patternS, firstCtx, posv & 0x3ff, log2n, chroma);
        } else { cur = (kk <= top) ? rqCur[sp] : 0; sigIdx = rqCtxSig[sp]; }
        HM_LVK(sCur, k) = cur; HM_LVK(sDec, k) = dec; HM_LVK(sLvl, k) = rqLvl[sp]; HM_LVK(sNeg, k) = (posv >> 15) & 1;
        const int d = HM_LV_GATHER(tSig, sigIdx * 2 + 1) - HM_LV_GATHER(tSig, sigIdx * 2);
        HM_LVK(sDelta, k) = (sp == lastScanPos) ? 0 : d;
        HM_BALLOT(nzM, k, cur != 0 && k < 16);
        HM_BALLOT(oddM, k, (cur & 1) && k < 16);
      }
      const int lastNZ = nzM ? 63 - __builtin_clzll(nzM) : -1, firstNZ = nzM ? __builtin_ctzll(nzM) : 16;
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4) {
        const uint32_t signbit = HM_LV_GET(sCur, firstNZ) > 0 ? 0 : 1;
        if (signbit != (uint32_t)(__builtin_popcountll(oddM) & 1)) {
          const int64_t I64MAX = 0x7fffffffffffffffLL;
          int64_t minCostInc = I64MAX, curCost = I64MAX; int minK = -1, finalChange = 0, curChange = 0;
          // re-walk the group's decision-time state (contexts, Rice parameter, flag counters)
          int wC1 = 1, wC1Idx = 0, wC2Idx = 0, wGoR = 0; const int wSet = cgCtxSet[subSet];
          const int kStart = (lastCG == 1 ? lastNZ : 15);
          for (int k = top; k >= 0; --k) {
            const uint32_t dec = (uint32_t)HM_LV_GET(sDec, k);
            const int ctxOne = 4 * wSet + wC1, ctxSetD = wSet, goR = wGoR, c1I = wC1Idx, c2I = wC2Idx;
            { // advance the walk past this position (same updates as the decision chain)
              const uint32_t baseLevel = (wC1Idx < 8) ? (2 + (wC2Idx < 1)) : 1;
              if (dec >= baseLevel && dec > (3u << wGoR)) wGoR = wGoR + 1 < 4 ? wGoR + 1 : 4;
              if (dec >= 1) wC1Idx++;
              if (dec > 1) { wC1 = 0; wC2Idx++; }
              else if (wC1 < 3 && wC1 > 0 && dec) wC1++;
            }
            if (k > kStart) continue;
            const int32_t lvlD = HM_LV_GET(sLvl, k);
            const int32_t deltaU = (int32_t)((lvlD - (int32_t)(dec << qBits)) >> (qBits - 8));
            const int sigRateDelta = HM_LV_GET(sDelta, k);
            int rateIncUp, rateIncDown = 0;
            if (dec > 0) {
              const int rateNow = ic_rate(tOne, dec, ctxOne, ctxSetD, goR, c1I, c2I);
              rateIncUp = ic_rate(tOne, dec + 1, ctxOne, ctxSetD, goR, c1I, c2I) - rateNow;
              rateIncDown = ic_rate(tOne, dec - 1, ctxOne, ctxSetD, goR, c1I, c2I) - rateNow;
            } else rateIncUp = HM_LV_GET(tOne, ctxOne * 2);
            const int dv = HM_LV_GET(sCur, k);
            if (dv != 0) {
              const int64_t costUp = rdFactor * (-deltaU) + rateIncUp;
              int64_t costDown = rdFactor * (deltaU) + rateIncDown - ((hm_abs(dv) == 1) ? sigRateDelta : 0);
              if (lastCG == 1 && lastNZ == k && hm_abs(dv) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else { curChange = -1; if (k == firstNZ && hm_abs(dv) == 1) curCost = I64MAX; else curCost = costDown; }
            } else {
              curCost = rdFactor * (-(hm_abs(deltaU))) + (1 << 15) + rateIncUp + sigRateDelta;
              curChange = 1;
              if (k < firstNZ) { const uint32_t thissign = (uint32_t)HM_LV_GET(sNeg, k); if (thissign != signbit) curCost = I64MAX; }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minK = k; }
          }
          if (minK >= 0) {
            const int mv = HM_LV_GET(sCur, minK);
            if (mv == 32767 || mv == -32768) finalChange = -1;
            const int nv = HM_LV_GET(sNeg, minK) ? mv - finalChange : mv + finalChange;
            if (big) HM_LV_SET(sCur, minK, nv); else rqCur[subPos + minK] = (int16_t)nv;
          }
        }
      }
      if (big) { HM_WAVE_FOR(k) { if (k < 16) rqDec[subPos + k] = (uint16_t)(int16_t)HM_LVK(sCur, k); } }   // this group's final levels take its decisions' place
      if (lastCG == 1) lastCG = 0;
    }
  }
  HM_SYNC();
  HM_PAR_FOR(sp, lastScanPos + 1) {
    int v;
    if (!big) v = rqCur[sp];
    else if (absSum >= 2) v = (int16_t)rqDec[sp];
    else { const int lv = (sp < bestLastIdxP1 && ((cgMask >> scanCG[sp >> 4]) & 1)) ? (int)rqDec[sp] : 0; v = (rqPos[sp] & 0x8000) ? -lv : lv; }
    dst[rqPos[sp] & 0x3ff] = v;
  }
  HM_SYNC();
  return absSum;
}

// ------------------------------------------------------------------------------------------------
// syntax element coding on the estimator (TEncSbac.cpp)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// The estimator in registers.  One Cabac (163 context states, 4 per dword) fits one lane variable; the entropy
// table (2 x 64 dwords) and the LPS transition table (128 bytes = 32 dwords) sit in three more.  A bin then costs
// a few v_readlane + scalar instructions and no memory access.  Functions that code many bins load the state
// from its LDS home at entry and store it back at exit.
// ------------------------------------------------------------------------------------------------
struct CabacR { HM_LV(int32_t, st); HM_LV(int32_t, eb0); HM_LV(int32_t, eb1); HM_LV(int32_t, lps); uint64_t frac; };
HM_DEV inline void cabr_load(const Shared *e, CabacR &r, const Cabac *c)
{
  HM_WAVE_FOR(k) {
    HM_LVK(r.st, k) = k < (int)(sizeof(c->s) / 4) ? ((const int32_t *)c->s)[k] : 0;
    HM_LVK(r.eb0, k) = HM_LT()->ebits[k]; HM_LVK(r.eb1, k) = HM_LT()->ebits[64 + k];
    HM_LVK(r.lps, k) = ((const int32_t *)HM_NEXT_LPS)[k & 31];
  }
  r.frac = c->frac;
}
HM_DEV inline void cabr_store(const CabacR &r, Cabac *c)
{
  HM_WAVE_FOR(k) { if (k < (int)(sizeof(c->s) / 4)) ((int32_t *)c->s)[k] = HM_LVK(r.st, k); }
  c->frac = r.frac;
  HM_SYNC();
}
HM_DEV inline void enc_bin(const Shared *e, CabacR *r, int ctx, int bin)
{
  const int w = HM_LV_GET(r->st, ctx >> 2), sh = (ctx & 3) * 8, s = (w >> sh) & 0xff, i = s ^ bin;
  const int b0 = HM_LV_GET(r->eb0, i & 63), b1 = HM_LV_GET(r->eb1, i & 63);
  r->frac += (uint64_t)(uint32_t)(i < 64 ? b0 : b1);
  const int lw = HM_LV_GET(r->lps, s >> 2);
  const int ns = (bin == (s & 1)) ? (s < 124 ? s + 2 : s) : ((lw >> ((s & 3) * 8)) & 0xff);
  HM_LV_SET(r->st, ctx >> 2, (w & ~(0xff << sh)) | (ns << sh));
}
HM_DEV inline void enc_ep(CabacR *r, int n) { r->frac += (uint64_t)32768 * (uint64_t)n; }
// n bypass bins carrying a value (first bin = MSB of val): the estimators only count them
HM_DEV inline void enc_epv(Cabac *c, uint32_t val, int n) { (void)val; enc_ep(c, n); }
HM_DEV inline void enc_epv(CabacR *r, uint32_t val, int n) { (void)val; enc_ep(r, n); }
#include "hm355_bits.h"
HM_DEV inline int hm_min_in_group(int g) { return g < 4 ? g : ((2 + (g & 1)) << ((g >> 1) - 1)); }    // g_uiMinInGroup, TComRom.cpp

// TEncSbac::codeCoeffNxN, TEncSbac.cpp:1172-1525 (+codeLastSignificantXY :1106, xWriteCoefRemainExGolomb :337)
// Wavefront form: the coefficients are staged into LDS in scan order lane-parallel (last significant position by
// wave-max, coefficient-group flags by ballot); per coefficient group the 16 levels, their significance contexts
// and the >0 / >1 / >2 masks are produced lane-parallel, and the bins are coded by the serial context chain on
// register-resident data (CabacR).
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
#define HM_BPROF(id) do { if (real) { const unsigned long long t_ = __builtin_readcyclecounter(); e->prof[id] += t_ - profT1; e->profCnt[id] += 1; profT1 = t_; } } while (0)
#else
#define HM_BPROF(id) ((void)0)
#endif
template <class C> HM_DEV HM_NOINLINE void code_coeff_nxn(Shared *e, C *c, const TCoeff *coef, int n, int comp, int scanType, int tskipFlag)
{
  HM_ENTRY(e); n = HM_UNI(n); comp = HM_UNI(comp); scanType = HM_UNI(scanType); tskipFlag = HM_UNI(tskipFlag); c = hm_uni_ptr(c); HM_ASSUME_LDS(c); coef = hm_uni_ptr(coef);
  const int chroma = comp != 0, log2n = hm_log2(n), wg = n >> 2;
  const bool real = EngOf<C>::REAL != 0;          // the arithmetic coder also needs the values of the bypass bins
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  const unsigned long long profT0 = __builtin_readcyclecounter();
#endif
  typename EngOf<C>::R r; cabr_load(e, r, c);
  if (n == 4) enc_bin(e, &r, C_TSKIP + chroma, tskipFlag);              // codeTransformSkipFlags, TEncSbac.cpp:988
  const uint16_t *scan = e->tab->scan[scanType][log2n - 2], *scanCG = e->tab->scanCG[scanType][log2n - 2];
  // levels and raster positions in scan order: 32x32 blocks stage them in the idle transform buffer (2 x 2 KB of bufA's 4.1 KB)
  int16_t *lv = (n == 32) ? (int16_t *)e->bufA : e->u.rq.cur; uint8_t *cgFlag = e->u.rq.cgFlag;
  uint16_t *sposArr = (n == 32) ? (uint16_t *)e->bufA + 1024 : e->u.rq.pos;
  HM_PAR_FOR(i, 64) cgFlag[i] = 0;
  HM_SYNC();
  int lastLocal = -1;
  HM_PAR_FOR(sp, n * n) {
    const int blkPos = scan[sp]; const TCoeff v = coef[blkPos];
    lv[sp] = (int16_t)v; sposArr[sp] = (uint16_t)blkPos;
    if (v != 0) { if (sp > lastLocal) lastLocal = sp; const int py = blkPos >> log2n, px = blkPos - (py << log2n); cgFlag[wg * (py >> 2) + (px >> 2)] = 1; }
  }
  const int scanPosLast = hm_wave_max_i(lastLocal);
  HM_SYNC();
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  if (real) { e->prof[PR_ADI] += __builtin_readcyclecounter() - profT0; e->profCnt[PR_ADI] += 1; }
  unsigned long long profT1 = __builtin_readcyclecounter();
#endif
  uint64_t cgMask = 0;                    // coefficient-group flags, bit = raster position of the group
  HM_LV(int32_t, vScanCG);                // raster position of the group at scan index = lane
  HM_WAVE_FOR(k) { HM_BALLOT(cgMask, k, cgFlag[k] != 0); HM_LVK(vScanCG, k) = k < wg * wg ? scanCG[k] : 0; }
  const int posLast = sposArr[scanPosLast];
  {
    int py = posLast >> log2n, px = posLast - (py << log2n);
    if (scanType == SCAN_VER) { const int t = px; px = py; py = t; }
    const int gx = hm_group_idx(px), gy = hm_group_idx(py), gmax = hm_group_idx(n - 1);
    int off, shift; last_ctx_params(chroma, n, &off, &shift);
    const int bxc = C_LASTX + (chroma ? 15 : 0) + off, byc = C_LASTY + (chroma ? 15 : 0) + off;
    int k;
    for (k = 0; k < gx; k++) enc_bin(e, &r, bxc + (k >> shift), 1);
    if (gx < gmax) enc_bin(e, &r, bxc + (k >> shift), 0);
    for (k = 0; k < gy; k++) enc_bin(e, &r, byc + (k >> shift), 1);
    if (gy < gmax) enc_bin(e, &r, byc + (k >> shift), 0);
    if (gx > 3) enc_epv(&r, (uint32_t)(px - hm_min_in_group(gx)), (gx - 2) >> 1);
    if (gy > 3) enc_epv(&r, (uint32_t)(py - hm_min_in_group(gy)), (gy - 2) >> 1);
  }
  HM_BPROF(PR_INV);       // last position
  const int firstCtx = first_sig_ctx(n, scanType, chroma), sigOff = C_SIG + (chroma ? 28 : 0);
  const int lastScanSet = scanPosLast >> 4;
  int c1 = 1;
  for (int subSet = lastScanSet; subSet >= 0; subSet--) {
    const int subPos = subSet << 4, isLastSet = subSet == lastScanSet;
    const int cgBlkPos = HM_LV_GET(vScanCG, subSet), cgy = cgBlkPos >> (log2n - 2), cgx = cgBlkPos & (wg - 1);
    const int sigRight = (cgx < wg - 1) ? (int)((cgMask >> (cgBlkPos + 1)) & 1) : 0;
    const int sigLower = (cgy < wg - 1) ? (int)((cgMask >> (cgBlkPos + wg)) & 1) : 0;
    if (isLastSet || subSet == 0) cgMask |= 1ull << cgBlkPos;
    else enc_bin(e, &r, C_SIG_CG + (chroma ? 2 : 0) + ((sigRight + sigLower) != 0), (int)((cgMask >> cgBlkPos) & 1));
    if (!((cgMask >> cgBlkPos) & 1)) continue;
    const int pattern = wg <= 1 ? 0 : sigRight + (sigLower << 1);
    const int top = isLastSet ? (scanPosLast & 15) : 15;            // highest coded position of this group
    HM_LV(int32_t, vCtx); HM_LV(int32_t, vAbs);
    uint64_t nz = 0, g1 = 0, g2 = 0, ng = 0;
    HM_WAVE_FOR(k) {
      const int kk = k & 15, sp = subPos + kk;
      const int a = (kk <= top) ? hm_abs((int)lv[sp]) : 0;
      if (real) HM_BALLOT(ng, k, kk <= top && lv[sp] < 0 && k < 16);
      HM_LVK(vAbs, k) = a;
      HM_LVK(vCtx, k) = sig_ctx_inc(pattern, firstCtx, sposArr[sp], log2n, chroma);
      HM_BALLOT(nz, k, a != 0 && k < 16); HM_BALLOT(g1, k, a > 1 && k < 16); HM_BALLOT(g2, k, a > 2 && k < 16);
    }
    HM_BPROF(PR_SATD35);  // per-group preparation
    { // significance flags; the last significant coefficient itself is implied
      int seen = isLastSet ? 1 : 0;
      for (int p = isLastSet ? top - 1 : 15; p >= 0; p--) {
        const int sig = (int)((nz >> p) & 1);
        if (p > 0 || subSet == 0 || seen) enc_bin(e, &r, sigOff + HM_LV_GET(vCtx, p), sig);
        seen += sig;
      }
    }
    HM_BPROF(PR_TUBLK);   // significance flags
    const int numNonZero = __builtin_popcountll(nz);
    if (numNonZero > 0) {
      const int lastNZ = 63 - __builtin_clzll(nz), firstNZ = __builtin_ctzll(nz);
      const int signHidden = (lastNZ - firstNZ >= 4);
      const int ctxSet = ctx_set_index(chroma, subSet, c1 == 0);
      c1 = 1;
      int firstC2 = -1, escape = 0, idx = 0;
      for (uint64_t m = nz; m && idx < 8; idx++) {
        const int p = 63 - __builtin_clzll(m); m &= ~(1ull << p);
        const int sym = (int)((g1 >> p) & 1);
        enc_bin(e, &r, C_ONE + 4 * ctxSet + c1, sym);
        if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = p; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && firstC2 != -1) { const int sym = (int)((g2 >> firstC2) & 1); enc_bin(e, &r, C_ABS + ctxSet, sym); if (sym) escape = 1; }
      escape = escape || (numNonZero > 8);
      uint32_t signs = 0;                 // coeff_sign_flag of the non-zero levels, highest scan position first; the hidden one is the last
      if (real) { for (uint64_t m = nz; m;) { const int p = 63 - __builtin_clzll(m); m &= ~(1ull << p); signs = (signs << 1) | (uint32_t)((ng >> p) & 1); } if (signHidden) signs >>= 1; }
      enc_epv(&r, signs, signHidden ? numNonZero - 1 : numNonZero);
      if (escape) {
        int firstCoeff2 = 1; uint32_t goRice = 0; idx = 0;
        for (uint64_t m = nz; m; idx++) {
          const int p = 63 - __builtin_clzll(m); m &= ~(1ull << p);
          const int a = HM_LV_GET(vAbs, p);
          const int baseLevel = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (a >= baseLevel) {
            uint32_t sym = (uint32_t)(a - baseLevel);
            // xWriteCoefRemainExGolomb :337: unary prefix + rice suffix, or the escape (prefix of 4+ ones, exp-golomb suffix)
            if (sym < (3u << goRice)) { const uint32_t len = sym >> goRice; enc_epv(&r, (((1u << (len + 1)) - 2) << goRice) | (sym & ((1u << goRice) - 1)), (int)(len + 1 + goRice)); }
            else { uint32_t len = goRice; sym -= (3u << goRice); while (sym >= (1u << len)) sym -= (1u << (len++));
                   if (real) { enc_epv(&r, (1u << (3 + len + 1 - goRice)) - 2, (int)(3 + len + 1 - goRice)); enc_epv(&r, sym, (int)len); }
                   else enc_epv(&r, 0, (int)(3 + len + 1 - goRice + len)); }
            if ((uint32_t)a > (3u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
          }
          if (a >= 2) firstCoeff2 = 0;
        }
      }
    }
    HM_BPROF(PR_CHROMA);  // levels, signs, remaining
  }
  cabr_store(r, c);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  if (real) { e->prof[PR_FWD] += __builtin_readcyclecounter() - profT0; e->profCnt[PR_FWD] += 1; }
#endif
}


// TEncSbac::codeIntraDirLumaAng, TEncSbac.cpp:636-690
template <class C> HM_DEV inline void code_intra_dir_luma(Shared *e, C *c, int z, int multiple)
{
  const CtuMeta *m = (&e->meta);
  const int partNum = multiple ? (m->part[z] == SIZE_NxN ? 4 : 1) : 1;
  const int partOffset = (256 >> (m->depth[z] << 1)) >> 2;
  int predIdx[4] = {-1, -1, -1, -1};
  for (int j = 0; j < partNum; j++) {
    int preds[3];
    const int dir = m->dirL[z + partOffset * j];
    if (!multiple && z == e->mpmZ) { preds[0] = e->mpmPreds[0]; preds[1] = e->mpmPreds[1]; preds[2] = e->mpmPreds[2]; }
    else intra_dir_predictor(e, z + partOffset * j, preds);
    for (int i = 0; i < 3; i++) if (dir == preds[i]) predIdx[j] = i;
    enc_bin(e, c, C_INTRA_LUMA, predIdx[j] != -1);
  }
  for (int j = 0; j < partNum; j++) {
    if (predIdx[j] != -1) { if (predIdx[j]) enc_epv(c, 2u | (uint32_t)(predIdx[j] - 1), 2); else enc_epv(c, 0, 1); }
    else {
      uint32_t rem = 0;                   // rem_intra_luma_pred_mode: the mode minus the number of smaller candidates
      if (EngOf<C>::REAL) {
        int p[3]; intra_dir_predictor(e, z + partOffset * j, p);
        int d = m->dirL[z + partOffset * j];
        rem = (uint32_t)(d - (d > p[0]) - (d > p[1]) - (d > p[2]));
      }
      enc_epv(c, rem, 5);
    }
  }
}
// TEncSbac::codeIntraDirChroma, TEncSbac.cpp:692-718
template <class C> HM_DEV inline void code_intra_dir_chroma(Shared *e, C *c, int z)
{
  if ((&e->meta)->dirC[z] == DM_CHROMA_IDX) enc_bin(e, c, C_CHROMA_PRED, 0);
  else {
    enc_bin(e, c, C_CHROMA_PRED, 1);
    uint32_t idx = 0;
    if (EngOf<C>::REAL) { // position in getAllowedChromaDir's list (TComDataCU.cpp:1486): planar, vertical, horizontal, DC, the one equal to the luma mode replaced by 34
      const int luma = (&e->meta)->dirL[z], dc = (&e->meta)->dirC[z];
      const int list[4] = { PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX };
      for (int i = 3; i >= 0; i--) if ((list[i] == luma ? 34 : list[i]) == dc) idx = (uint32_t)i;
    }
    enc_epv(c, idx, 2);
  }
}
// TEncSbac::codeQtCbf, TEncSbac.cpp:911-960 (square TUs)
template <class C> HM_DEV inline void code_qt_cbf(Shared *e, C *c, const TU *t, int comp, int lowestLevel)
{
  const int z = t->cuZ + (comp ? t->cRelZ : t->relZ);
  const int ctx = comp ? t->trDepth : (t->trDepth == 0 ? 1 : 0);
  const int width = comp ? (1 << (t->log2 - 1)) : (1 << t->log2);
  const int canQuadSplit = width >= 8;
  const int lowestTUDepth = t->trDepth + ((!lowestLevel && !canQuadSplit) ? 1 : 0);
  enc_bin(e, c, C_QT_CBF + (comp ? 5 : 0) + ctx, ((&e->meta)->cbf[comp][z] >> lowestTUDepth) & 1);
}

HM_DEV inline int tr_min_size_in_cu(int cuLog2, int nxn)
{ // TComDataCU::getQuadtreeTULog2MinSizeInCU, TComDataCU.cpp:1618-1643 (max depth intra 3, TU log2 2..5)
  if (cuLog2 < 2 + 3 - 1 + nxn) return 2;
  const int v = cuLog2 - (3 - 1 + nxn);
  return v > 5 ? 5 : v;
}
// whether a transform_split flag is coded at this node (TEncSearch.cpp:869-888 / TEncEntropy.cpp:258-291)
HM_DEV inline int codes_subdiv_flag(const CtuMeta *m, const TU *t)
{
  const int nxn = m->part[t->cuZ] == SIZE_NxN;
  if (nxn && t->trDepth == 0) return 0;
  if (t->log2 > 5) return 0;
  if (t->log2 == 2) return 0;
  if (t->log2 == tr_min_size_in_cu(6 - t->cuDepth, nxn)) return 0;
  return 1;
}

// Pre-order walk of the residual quadtree with an explicit stack (no device recursion).
// The callbacks of the reference's recursive functions become phases of one loop.
HM_DEV inline void walk_begin(TuWalk *w, const TU *root) { w->node[0] = *root; w->next[0] = -1; w->sp = 0; }

// xEncSubdivCbfQT, TEncSearch.cpp:856-921
template <class C> HM_DEV inline void enc_subdiv_cbf_qt(Shared *e, C *c, const TU *root, int bLuma, int bChroma)
{
  const CtuMeta *m = (&e->meta);
  TuWalk &w = e->walkInner; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    const int subdiv = m->tr[z] > t->trDepth;
    if (w.next[w.sp] < 0) { // first visit
      if (bLuma && codes_subdiv_flag(m, t)) enc_bin(e, c, C_SUBDIV + (5 - t->log2), subdiv);
      if (bChroma)
        for (int comp = 1; comp < 3; comp++)
          if (t->cCodeAll && (t->trDepth == 0 || ((m->cbf[comp][z] >> (t->trDepth - 1)) & 1)))
            code_qt_cbf(e, c, t, comp, subdiv == 0);
      if (!subdiv) { if (bLuma) code_qt_cbf(e, c, t, 0, 1); w.sp--; continue; }
      w.next[w.sp] = 0;
    }
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
  }
}
// xEncCoeffQT, TEncSearch.cpp:926-960 (coefficients from the QT layer buffers)
HM_DEV inline void enc_coeff_qt(Shared *e, const TU *root, int comp)
{
  const CtuMeta *m = (&e->meta);
  TuWalk &w = e->walkInner; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    if (m->tr[z] > t->trDepth) {
      if (w.next[w.sp] < 0) w.next[w.sp] = 0;
      if (w.next[w.sp] == 4) { w.sp--; continue; }
      const int s = w.next[w.sp]++;
      w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
      continue;
    }
    if (!(comp && !t->cW) && ((m->cbf[comp][z] >> t->trDepth) & 1)) {   // TEncEntropy::encodeCoeffNxN, TEncEntropy.cpp:683
      const int layer = 5 - t->log2;
      const int n = comp ? t->cW : (1 << t->log2);
      const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
      const TCoeff *coef = e->ws->qtCoef[layer] + HM_PLANE_OFF(comp) + (comp ? t->cOff : z * 16);
      code_coeff_nxn(e, &e->cur, coef, n, comp, coef_scan_idx(m, zc, n, comp), m->ts[comp][zc]);
    }
    w.sp--;
  }
}
// xEncIntraHeader, TEncSearch.cpp:965-1032 (I slice, no PCM)
template <class C> HM_DEV inline void enc_intra_header(Shared *e, C *c, const TU *t, int bLuma, int bChroma)
{
  const CtuMeta *m = (&e->meta); const int relZ = t->relZ;
  if (bLuma) {
    if (relZ == 0 && e->im) { code_skip_flag(e, c, t->cuZ); enc_bin(e, c, C_PRED_MODE, 1); }   // P / B slices: skip flag + pred mode, TEncSearch.cpp:975-984
    if (relZ == 0 && t->cuDepth == 3) enc_bin(e, c, C_PART, m->part[t->cuZ] == SIZE_2Nx2N);
    if (m->part[t->cuZ] == SIZE_2Nx2N) { if (relZ == 0) code_intra_dir_luma(e, c, t->cuZ, 0); }
    else { const int q = t->cuParts >> 2; if (t->trDepth > 0 && (relZ & (q - 1)) == 0) code_intra_dir_luma(e, c, t->cuZ + relZ, 0); }
  }
  if (bChroma && relZ == 0) code_intra_dir_chroma(e, c, t->cuZ + relZ);
}
// xGetIntraBitsQT, TEncSearch.cpp:1038-1060
HM_DEV HM_NOINLINE uint32_t intra_bits_qt(Shared *e, TU tv, int bLuma, int bChroma)
{
  HM_ENTRY(e); bLuma = HM_UNI(bLuma); bChroma = HM_UNI(bChroma); tv = hm_uni_struct(tv);
  const TU *t = &tv;
  HM_PROF_BEGIN(e, PR_BITS);
  {
    CabacR r; cabr_load(e, r, &e->cur);
    r.frac &= 32767;                       // resetBits, TEncBinCoderCABAC.cpp:161
    enc_intra_header(e, &r, t, bLuma, bChroma);
    enc_subdiv_cbf_qt(e, &r, t, bLuma, bChroma);
    cabr_store(r, &e->cur);
  }
  if (bLuma) enc_coeff_qt(e, t, 0);
  if (bChroma) { enc_coeff_qt(e, t, 1); enc_coeff_qt(e, t, 2); }
  HM_PROF_END(e, PR_BITS);
  return num_bits(&e->cur);
}

// ------------------------------------------------------------------------------------------------
// one TU: predict, transform, RDOQ, reconstruct (TEncSearch::xIntraCodingTUBlock :1074-1357)
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE uint32_t intra_coding_tu_block(Shared *e, TU tv, int comp, int save1load2)
{
  HM_ENTRY(e); comp = HM_UNI(comp); save1load2 = HM_UNI(save1load2); tv = hm_uni_struct(tv);
  const TU *t = &tv;
  CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  if (comp && !t->cW) return 0;
  const int n = comp ? t->cW : (1 << t->log2), l2 = hm_log2(n);
  const int relZ = comp ? t->cRelZ : t->relZ, z = t->cuZ + relZ;
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
  const int st = HM_PLANE_STRIDE(comp), po = HM_PLANE_OFF(comp);
  const int layer = 5 - t->log2, parts = comp ? t->cParts : t->parts;
  const int ps = e->stride[comp], bitDepth = e->bitDepth;
  const Pel *org = e->fb.org[comp] + (e->ctuY * st + by) * ps + e->ctuX * st + bx;
  Pel *recPic = e->fb.rec[comp] + (e->ctuY * st + by) * ps + e->ctuX * st + bx;
  Pel *pred = ws->pred + po + by * st + bx, *resi = ws->resi + po + by * st + bx;
  Pel *recQt = ws->qtRec[layer] + po + by * st + bx;
  TCoeff *coef = ws->qtCoef[layer] + po + (comp ? t->cOff : z * 16);
  const int tskip = m->ts[comp][z];
  int mode = comp ? m->dirC[z] : m->dirL[z];
  if (comp && mode == DM_CHROMA_IDX) mode = m->dirL[z & ~3];
  if (save1load2 != 2) {
    const int filt = use_filtered_refs(comp, mode, n);
    const int r = hm_z2r(z);
    { HM_PROF_BEGIN(e, PR_ADI); init_adi_pattern(e, comp, e->ctuX * st + bx, e->ctuY * st + by, n, e->ctuX * 16 + (r & 15), e->ctuY * 16 + (r >> 4), comp ? n / 2 : n / 4, filt); HM_PROF_END(e, PR_ADI); }
    { HM_PROF_BEGIN(e, PR_PRED); pred_intra(e, comp, mode, n, filt, pred, st); HM_PROF_END(e, PR_PRED); }
    if (save1load2 == 1) { HM_PAR_FOR(i, 16) e->ws->tsPred[comp][i] = pred[(i >> 2) * st + (i & 3)]; HM_SYNC(); }
  } else { HM_PAR_FOR(i, 16) pred[(i >> 2) * st + (i & 3)] = e->ws->tsPred[comp][i]; HM_SYNC(); }
  if (comp == 0) par_set8(m->tr + z, t->trDepth, parts);             // setTrIdxSubParts, TEncSearch.cpp:1229
  // residual straight into the LDS transform buffer (and the scratch plane, as the reference keeps it)
  const int tshift = 15 - bitDepth - l2;
  HM_PAR_FOR(i, n * n) {
    const int y = i >> l2, x = i & (n - 1);
    const int r = org[y * ps + x] - pred[y * st + x];
    resi[y * st + x] = (Pel)r;
    e->bufA[y * HM_TSTRIDE + x] = tskip ? (r << tshift) : r;         // xTransformSkip, TComTrQuant.cpp:1874
  }
  HM_SYNC();
  { HM_PROF_BEGIN(e, PR_FWD); if (!tskip) fwd_transform(e, n, comp == 0 && n == 4, bitDepth); HM_PROF_END(e, PR_FWD); }      // TComTrQuant::xT, :1805
  const int cbfCtx = comp ? 5 + t->trDepth : (t->trDepth == 0 ? 1 : 0);
  HM_PROF_BEGIN(e, PR_RDOQ);
  const int absSum = rdoq_is_empty(e, n, comp) ? 0 : (int)HM_UCALL(rdoq(e, coef, n, comp, coef_scan_idx(m, z, n, comp), cbfCtx));
  HM_PROF_END(e, PR_RDOQ);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  { const int pid = l2 == 2 ? (absSum ? 13 : 7) : (l2 == 3 ? 14 : (l2 == 4 ? 15 : 43)); e->prof[pid] += __builtin_readcyclecounter() - prof_t0_PR_RDOQ; e->profCnt[pid] += 1; }
#endif
  par_set8(m->cbf[comp] + z, (absSum > 0 ? 1 : 0) << t->trDepth, parts);   // setCbfPartRange, TComTrQuant.cpp:1419
  HM_PROF_BEGIN(e, PR_INV);
  if (absSum > 0) { // invTransformNxN, TComTrQuant.cpp:1423-1545; xDeQuant (flat), :1276-1312
    const int rightShift = 6 - (tshift + e->fb.qpPer[comp != 0]);
    const int scale = HM_INV_QUANT_SCALES[e->fb.qpRem[comp != 0]];
    int tgt = 25 + rightShift; if (tgt > 16) tgt = 16;
    const int imin = -(1 << (tgt - 1)), imax = (1 << (tgt - 1)) - 1;
    HM_PAR_FOR(i, n * n) {
      const int y = i >> l2, x = i & (n - 1);
      const int c = hm_clip3(imin, imax, coef[i]);
      int v;
      if (rightShift > 0) v = (c * scale + (1 << (rightShift - 1))) >> rightShift;
      else v = (int)((unsigned)(c * scale) << (-rightShift));
      e->bufA[y * HM_TSTRIDE + x] = hm_clip3(-32768, 32767, v);
    }
    HM_SYNC();
    if (!tskip) inv_transform(e, n, comp == 0 && n == 4, bitDepth);
    const int off = tshift == 0 ? 0 : (1 << (tshift - 1));
    HM_PAR_FOR(i, n * n) {
      const int y = i >> l2, x = i & (n - 1);
      const int v = e->bufA[y * HM_TSTRIDE + x];
      resi[y * st + x] = (Pel)(tskip ? ((v + off) >> tshift) : v);      // xITransformSkip, :1920
    }
  } else {
    HM_PAR_FOR(i, n * n) { const int y = i >> l2, x = i & (n - 1); coef[i] = 0; resi[y * st + x] = 0; }
  }
  HM_SYNC();
  HM_PROF_END(e, PR_INV);
  const int maxv = (1 << bitDepth) - 1, shiftSse = (bitDepth - 8) << 1;
  uint32_t sse = 0;
  HM_PAR_FOR(i, n * n) { // reconstruction + SSE fused (TEncSearch.cpp:1338-1356)
    const int y = i >> l2, x = i & (n - 1);
    const int r = hm_clip3(0, maxv, pred[y * st + x] + resi[y * st + x]);
    pred[y * st + x] = (Pel)r; recQt[y * st + x] = (Pel)r; recPic[y * ps + x] = (Pel)r;   // piReco aliases piPred
    const int d = org[y * ps + x] - r; sse += (uint32_t)((d * d) >> shiftSse);
  }
  uint32_t d = hm_wave_sum(sse);
  HM_SYNC();
  if (comp) d = (uint32_t)(e->fb.chromaWeight * (double)d);            // getDistPart, TComRdCost.cpp:447-450
  return d;
}

// xStoreIntraResultQT / xLoadIntraResultQT, TEncSearch.cpp:1790-1880 (4x4 blocks only: transform-skip trials)
HM_DEV inline void store_intra_result_qt(Shared *e, const TU *t, int comp)
{
  const int st = HM_PLANE_STRIDE(comp), po = HM_PLANE_OFF(comp), layer = 5 - t->log2;
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
  const TCoeff *coef = e->ws->qtCoef[layer] + po + (comp ? t->cOff : (t->cuZ + t->relZ) * 16);
  const Pel *rq = e->ws->qtRec[layer] + po + by * st + bx;
  HM_PAR_FOR(i, 16) { e->ws->tsCoef[comp][i] = coef[i]; e->ws->tsRec[comp][i] = rq[(i >> 2) * st + (i & 3)]; }
  HM_SYNC();
}
HM_DEV inline void load_intra_result_qt(Shared *e, const TU *t, int comp)
{
  const int st = HM_PLANE_STRIDE(comp), po = HM_PLANE_OFF(comp), layer = 5 - t->log2, ps = e->stride[comp];
  const int bx = comp ? t->cx : t->x, by = comp ? t->cy : t->y;
  TCoeff *coef = e->ws->qtCoef[layer] + po + (comp ? t->cOff : (t->cuZ + t->relZ) * 16);
  Pel *rq = e->ws->qtRec[layer] + po + by * st + bx;
  Pel *recPic = e->fb.rec[comp] + (e->ctuY * st + by) * ps + e->ctuX * st + bx;
  HM_PAR_FOR(i, 16) { coef[i] = e->ws->tsCoef[comp][i]; const Pel v = e->ws->tsRec[comp][i]; rq[(i >> 2) * st + (i & 3)] = v; recPic[(i >> 2) * ps + (i & 3)] = v; }
  HM_SYNC();
}

// ------------------------------------------------------------------------------------------------
// luma residual quadtree (TEncSearch::xRecurIntraCodingQT :1364-1733, bLumaOnly), explicit stack
// ------------------------------------------------------------------------------------------------
// returns distortion through *distY and adds the RD cost to *rdCost, exactly like the recursive reference
HM_DEV HM_NOINLINE void simt4_luma_leaf(Shared *e, TU tv);      // hm355_simt4.h
HM_DEV HM_NOINLINE void simt8_luma_winner_as_single_tu(Shared *e, TU tv);      // hm355_simt8.h
HM_DEV HM_NOINLINE void recur_intra_coding_qt(Shared *e, TU rootv, int checkFirst)
{
  HM_ENTRY(e); checkFirst = HM_UNI(checkFirst); rootv = hm_uni_struct(rootv);
  CtuMeta *m = (&e->meta);
  RqtFrame *fr = e->rqt; int sp = 0;
  uint32_t *retDist = e->rqtRetDist, *retBits = e->rqtRetBits; double *retCost = e->rqtRetCost;     // accumulators handed to each level by its parent
  fr[0].t = rootv; fr[0].phase = 0; retDist[0] = 0; retBits[0] = 0; retCost[0] = 0.0;
  while (sp >= 0) {
    RqtFrame *f = &fr[sp]; const TU *t = &f->t;
    const int z = t->cuZ + t->relZ, fullDepth = t->cuDepth + t->trDepth, log2 = t->log2;
    if (f->phase == 0) {
      const int nxn = m->part[t->cuZ] == SIZE_NxN;
      f->checkFull = log2 <= 5;
      f->checkSplit = log2 > tr_min_size_in_cu(6 - t->cuDepth, nxn);
      if (checkFirst && f->checkFull) f->checkSplit = 0;              // HHI_RQT_INTRA_SPEEDUP
      f->singleCost = HM_MAX_DOUBLE; f->singleDist = 0; f->singleCbf = 0; f->bestModeId = 0; f->singleBits = 0;
      const int checkTS = (log2 == 2) && (m->part[z] == SIZE_NxN);    // TransformSkip + TransformSkipFast
      if (f->checkFull) {
        if (checkTS) {
          cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
          for (int modeId = 0; modeId < 2; modeId++) {
            double costTmp;
            par_set8(m->ts[0] + z, modeId, t->parts);
            const uint32_t distTmp = HM_UCALL(intra_coding_tu_block(e, *t, 0, modeId == 0 ? 1 : 2));
            const uint32_t cbfTmp = (m->cbf[0][z] >> t->trDepth) & 1;
            if (modeId == 1 && cbfTmp == 0) costTmp = HM_MAX_DOUBLE;
            else { const uint32_t bits = HM_UCALL(intra_bits_qt(e, *t, 1, 0)); costTmp = calc_rd_cost(e, bits, distTmp); }
            if (costTmp < f->singleCost) {
              f->singleCost = costTmp; f->singleDist = distTmp; f->singleCbf = cbfTmp; f->bestModeId = (int8_t)modeId;
              if (modeId == 0) { store_intra_result_qt(e, t, 0); cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_TEMP_BEST)], &e->cur); }
            }
            if (modeId == 0) cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]);
          }
          par_set8(m->ts[0] + z, f->bestModeId, t->parts);
          if (f->bestModeId == 0) {
            load_intra_result_qt(e, t, 0);
            par_set8(m->cbf[0] + z, (int)(f->singleCbf << t->trDepth), t->parts);
            cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_TEMP_BEST)]);
          }
        } else if (log2 == 2 && e->P->fewWaves) {
          // a 4x4 leaf (it cannot split, so no snapshot is taken): evaluated and priced on one lane (simt4_luma_leaf) when the launch leaves
          // most of the device idle -- the per-lane form is the shorter dependency chain (one 4K picture: 14.1 s instead of 14.8 s) but issues
          // more instructions, which costs 2.5 % of throughput once every SIMD is shared by several searches (measured A/B on one device)
          par_set8(m->ts[0] + z, 0, t->parts);
          HM_PROF_BEGIN(e, PR_S4LEAF);
          simt4_luma_leaf(e, *t);
          HM_PROF_END(e, PR_S4LEAF);
          f->singleDist = e->outDistY; f->singleBits = e->outBits;
          f->singleCost = calc_rd_cost(e, e->outBits, f->singleDist);
        } else if (sp == 0 && e->s8Reuse == 1) {
          // closing pass of an 8x8 PU: its unsplit 8x8 TU is the evaluation the candidates-in-lanes first pass already made for the winner
          e->s8Reuse = 0;
          cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
          par_set8(m->ts[0] + z, 0, t->parts);
          par_set8(m->tr + z, t->trDepth, t->parts);
          simt8_luma_winner_as_single_tu(e, *t);
          f->singleDist = e->outDistY; f->singleCbf = e->outDist;
          par_set8(m->cbf[0] + z, (int)(f->singleCbf << t->trDepth), t->parts);
          f->singleCost = calc_rd_cost(e, e->outBits, f->singleDist);
        } else if (sp == 0 && e->s8Reuse == 2) {
          // closing pass of a 16x16 / 32x32 PU: its unsplit TU is the winner's first-pass evaluation (same snapshot, same mode), whose levels and
          // reconstruction est_intra_pred_qt kept (xSetIntraResultQT) and whose estimator state it parked in this depth's CI_QT_TRAFO_TEST slot
          e->s8Reuse = 0;
          cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
          par_set8(m->ts[0] + z, 0, t->parts);
          par_set8(m->tr + z, t->trDepth, t->parts);
          {
            const int n = 1 << log2, layer = 5 - log2;
            par_copy32(e->ws->qtCoef[layer] + z * 16, e->cc + z * 16, n * n);
            par_copy_blk(e->ws->qtRec[layer] + t->y * 64 + t->x, 64, e->ws->reco + t->y * 64 + t->x, 64, n);
          }
          f->singleCbf = e->ws->tmpCbf[0][0] & 1;
          par_set8(m->cbf[0] + z, (int)(f->singleCbf << t->trDepth), t->parts);
          f->singleDist = e->outDistY; f->singleCost = e->outRdCost;
          cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_TEST)]);
        } else {
          if (f->checkSplit) cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
          par_set8(m->ts[0] + z, 0, t->parts);
          f->singleDist = HM_UCALL(intra_coding_tu_block(e, *t, 0, 0));
          if (f->checkSplit) f->singleCbf = (m->cbf[0][z] >> t->trDepth) & 1;
          const uint32_t bits = HM_UCALL(intra_bits_qt(e, *t, 1, 0));
          f->singleCost = calc_rd_cost(e, bits, f->singleDist); f->singleBits = bits;
        }
      }
      if (!f->checkSplit) { retDist[sp] += f->singleDist; retBits[sp] += f->singleBits; retCost[sp] += f->singleCost; sp--; continue; }
      if (f->checkFull) { cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_TEST)], &e->cur); cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]); }
      else cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
      f->splitCost = 0.0; f->splitDist = 0; f->splitCbf = 0; f->child = 0; f->phase = 1;
      retDist[sp + 1] = 0; retBits[sp + 1] = 0; retCost[sp + 1] = 0.0;
    }
    if (f->phase == 1) {
      int splitLost = 0;
      if (f->child > 0) { // a child just returned
        const TU ch = tu_child(t, f->child - 1, 0);
        f->splitCbf |= (m->cbf[0][ch.cuZ + ch.relZ] >> ch.trDepth) & 1;
        // The split candidate is priced by coding the whole subtree again from this TU's snapshot (xGetIntraBitsQT, TEncSearch.cpp:1670).  A bin's price
        // depends only on the state of its own context, and each context sees the same bins in the same order in that recount as it did
        // while the children were searched one after the other (flags and coefficients use different contexts; bypass bins cost one bit
        // anywhere), so the recount is at least the sum of the children's prices - each known to within the fraction of a bit carried into
        // it.  Once that lower bound on the split cost is no cheaper than the unsplit TU, the remaining children and the recount are moot.
        const uint32_t lbBits = retBits[sp + 1] > (uint32_t)f->child ? retBits[sp + 1] - (uint32_t)f->child : 0;
        if (f->checkFull) splitLost = !(calc_rd_cost(e, lbBits, retDist[sp + 1]) < f->singleCost);
        else if (sp == 0 && !(calc_rd_cost(e, lbBits, retDist[1]) < e->outCost)) {
          // a 64x64 PU (four 32x32 TUs, no unsplit form): the same bound against the best candidate so far, which this one has to beat
          // (estIntraPredQT, TEncSearch.cpp:2490 / :2570) - it cannot, and nothing of it is kept
          e->outDistY = 0; e->outRdCost = HM_MAX_DOUBLE; return;
        }
      }
      if (!splitLost) {
        if (f->child < 4) {
          fr[sp + 1].t = tu_child(t, f->child, 0); fr[sp + 1].phase = 0; f->child++;
          sp++; continue;
        }
        f->splitDist = retDist[sp + 1];
        if (f->splitCbf) { HM_PAR_FOR(o, t->parts) m->cbf[0][z + o] |= (uint8_t)(1 << t->trDepth); HM_SYNC(); }
        cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]);
        const uint32_t splitBits = HM_UCALL(intra_bits_qt(e, *t, 1, 0));
        f->splitCost = calc_rd_cost(e, splitBits, f->splitDist);
        if (f->splitCost < f->singleCost) { retDist[sp] += f->splitDist; retBits[sp] += splitBits; retCost[sp] += f->splitCost; sp--; continue; }
      }
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_TEST)]);
      par_set8(m->tr + z, t->trDepth, t->parts);
      par_set8(m->cbf[0] + z, (int)(f->singleCbf << t->trDepth), t->parts);
      par_set8(m->ts[0] + z, f->bestModeId, t->parts);
      { // reconstruction of the unsplit TU back into the picture for the following blocks
        const int n = 1 << log2, layer = 5 - log2, ps = e->stride[0];
        par_copy_blk(e->fb.rec[0] + (e->ctuY * 64 + t->y) * ps + e->ctuX * 64 + t->x, ps, e->ws->qtRec[layer] + t->y * 64 + t->x, 64, n);
      }
      retDist[sp] += f->singleDist; retBits[sp] += f->singleBits; retCost[sp] += f->singleCost; sp--; continue;
    }
  }
  e->outDistY = retDist[0]; e->outRdCost = retCost[0];
}

// leaves of the decided residual quadtree inside [relZ0, relZ0+parts0): visit in z order
// xSetIntraResultQT, TEncSearch.cpp:1737-1788 (luma)
HM_DEV inline void set_intra_result_qt(Shared *e, const TU *root)
{
  const CtuMeta *m = (&e->meta);
  for (int rel = root->relZ; rel < root->relZ + root->parts;) {
    const int z = root->cuZ + rel, trd = m->tr[z];
    const int log2 = 6 - root->cuDepth - trd, n = 1 << log2, layer = 5 - log2;
    const int parts = root->cuParts >> (2 * trd) > 0 ? root->cuParts >> (2 * trd) : 1;
    const int r = hm_z2r(z), x = (r & 15) * 4, y = (r >> 4) * 4;
    par_copy32(e->cc + z * 16, e->ws->qtCoef[layer] + z * 16, n * n);
    par_copy_blk(e->ws->reco + y * 64 + x, 64, e->ws->qtRec[layer] + y * 64 + x, 64, n);
    rel += parts;
  }
}

#include "hm355_simt4.h"
#include "hm355_simt8.h"

// ------------------------------------------------------------------------------------------------
// luma mode decision of one CU (TEncSearch::estIntraPredQT :2289-2692)
// ------------------------------------------------------------------------------------------------
HM_DEV HM_NOINLINE uint32_t est_intra_pred_qt(Shared *e, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth);
  CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws;
  const int cuParts = 256 >> (2 * cuDepth);
  const int nxn = m->part[cuZ] == SIZE_NxN;
  const int numPU = nxn ? 4 : 1, puParts = cuParts / numPU;
  const int puLog2 = 6 - cuDepth - nxn, n = 1 << puLog2;
  const int bitDepth = e->bitDepth, ps = e->stride[0];
  uint32_t overallDistY = 0;
  const TU root = tu_root(e, cuZ, cuDepth);
  for (int pu = 0; pu < numPU; pu++) {
    const TU t = nxn ? tu_child(&root, pu, 0) : root;
    const int z = cuZ + t.relZ;
    int numModesForFullRD = HM_INTRA_MODE_NUM_FAST[puLog2 - 1];
    int32_t *rdModeList = e->rdModeList;
    { // SATD pre-selection over the 35 modes, :2360-2410
      const int r = hm_z2r(z);
      init_adi_pattern(e, 0, e->ctuX * 64 + t.x, e->ctuY * 64 + t.y, n, e->ctuX * 16 + (r & 15), e->ctuY * 16 + (r >> 4), n / 4, 1);
      const Pel *org = e->fb.org[0] + (e->ctuY * 64 + t.y) * ps + e->ctuX * 64 + t.x;
      Pel *pred = ws->pred + t.y * 64 + t.x;
      int preds[3];
      const int numMpm = intra_dir_predictor(e, z, preds);
      e->mpmZ = z; e->mpmNum = numMpm; e->mpmPreds[0] = preds[0]; e->mpmPreds[1] = preds[1]; e->mpmPreds[2] = preds[2];
      // xModeBitsIntra (TEncSearch.cpp:5456-5478) depends only on whether the mode is an MPM and which
      const uint64_t frac0 = e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)].frac & 32767;
      const uint8_t st0 = e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)].s[C_INTRA_LUMA];
      HM_PROF_BEGIN(e, PR_SATD35);
      if (n <= 16) satd_all_modes_small(e, org, ps, n);
      else {
        for (int mode = 0; mode < 35; mode++) {
          pred_intra(e, 0, mode, n, use_filtered_refs(0, mode, n), pred, 64);
          const uint32_t sad = dist_hads(org, ps, pred, 64, n, bitDepth);
          if (hm_lane() == 0) e->satd[mode] = sad;
        }
        HM_SYNC();
      }
      // xUpdateCandList (TEncSearch.cpp:5484-5505) over the 35 modes in order keeps the numModesForFullRD cheapest, a mode going behind
      // the equally cheap ones already listed: the list is the modes sorted by (cost, mode number), cut off.  One lane per mode
      // computes its cost and its rank among all of them.
      {
        HM_LV(double, vCost);
        HM_WAVE_FOR(k) {
          double cost = HM_MAX_DOUBLE;
          if (k < 35) {
            int predIdx = -1;
            for (int i = 0; i < 3; i++) if (k == preds[i]) predIdx = i;
            uint64_t fb = frac0 + (uint64_t)HM_LT()->ebits[st0 ^ (predIdx != -1)];
            fb += (uint64_t)32768 * (uint64_t)(predIdx == -1 ? 5 : (predIdx ? 2 : 1));
            const uint32_t modeBits = (uint32_t)(fb >> 15);
            cost = (double)e->satd[k] + (double)modeBits * e->fb.sqrtLambda;
          }
          HM_LVK(vCost, k) = cost;
        }
        HM_WAVE_FOR(k) {
          const double ck = HM_LVK(vCost, k);
          int rank = 0;
          for (int j = 0; j < 35; j++) { const double cj = HM_LV_GETD(vCost, j); rank += (cj < ck || (cj == ck && j < k)) ? 1 : 0; }
          if (k < 35 && rank < numModesForFullRD) rdModeList[rank] = k;
        }
        HM_SYNC();
      }
      HM_PROF_END(e, PR_SATD35);
      for (int j = 0; j < numMpm; j++) { // numCand = *piMode quirk, TEncSearch.cpp:2415-2420
        int included = 0;
        for (int i = 0; i < numModesForFullRD; i++) included |= (preds[j] == rdModeList[i]);
        if (!included) rdModeList[numModesForFullRD++] = preds[j];
      }
    }
    if (n == 4) {
      // 4x4 PUs (NxN): every candidate with and without transform skip, one per lane (hm355_simt4.h).  The reference's closing pass
      // with the best mode (:2566-2600) repeats the winner's evaluation unchanged -- a 4x4 TU cannot split -- and its strict "<" keeps
      // the first result, so it is not run again.
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      { HM_PROF_BEGIN(e, PR_S4L); simt4_luma_pu(e, t, numModesForFullRD); HM_PROF_END(e, PR_S4L); }
      overallDistY += e->outDistY;
      if (pu != numPU - 1)
        par_copy_blk(e->fb.rec[0] + (e->ctuY * 64 + t.y) * ps + e->ctuX * 64 + t.x, ps, ws->reco + t.y * 64 + t.x, 64, n);
      continue;
    }
    int bestPUMode = 0; uint32_t bestPUDistY = 0; double bestPUCost = HM_MAX_DOUBLE;
    int firstPass = 0;
    if (n == 8) {
      // 8x8 PU (2Nx2N CU of the smallest size): the first pass over the candidates only ranks them -- each is one unsplit 8x8 transform
      // block from the same snapshot -- so it runs with the candidates in lanes (hm355_simt8.h) and only the closing pass with the full
      // residual quadtree follows.  That pass starts with the very evaluation the winner had in the first pass, so its result is
      // the first-pass result or better, and is taken as the reference takes it (:2566-2600).
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      HM_PROF_BEGIN(e, PR_S8L);
      bestPUMode = HM_UCALL(simt8_luma_first_pass(e, t, numModesForFullRD));
      HM_PROF_END(e, PR_S8L);
      firstPass = numModesForFullRD;
      e->s8Reuse = 1;
    }
    for (int pass = firstPass; pass <= numModesForFullRD; pass++) {
      const int last = (pass == numModesForFullRD);
      const int orgMode = last ? bestPUMode : rdModeList[pass];
      par_set8(m->dirL + z, orgMode, puParts);
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      // 16x16 / 32x32 PUs: the closing pass takes its unsplit TU from the winner's first-pass evaluation instead of repeating it
      e->outCost = bestPUCost;               // what this candidate has to beat (recur_intra_coding_qt gives up on a 64x64 PU that cannot)
      if (last && (n == 16 || n == 32) && numModesForFullRD > 0) { e->s8Reuse = 2; e->outDistY = bestPUDistY; e->outRdCost = bestPUCost; }
      recur_intra_coding_qt(e, t, !last);
      const uint32_t puDistY = e->outDistY; const double puCost = e->outRdCost;
      if (puCost < bestPUCost) {
        bestPUMode = orgMode; bestPUDistY = puDistY; bestPUCost = puCost;
        set_intra_result_qt(e, &t);
        HM_PAR_FOR(i, puParts) { ws->tmpTr[i] = m->tr[z + i]; for (int c = 0; c < 3; c++) { ws->tmpCbf[c][i] = m->cbf[c][z + i]; ws->tmpTs[c][i] = m->ts[c][z + i]; } }
        HM_SYNC();
        if (!last && (n == 16 || n == 32)) cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_QT_TRAFO_TEST)], &e->cur);
      }
    }
    overallDistY += bestPUDistY;
    HM_PAR_FOR(i, puParts) { m->tr[z + i] = ws->tmpTr[i]; for (int c = 0; c < 3; c++) { m->cbf[c][z + i] = ws->tmpCbf[c][i]; m->ts[c][z + i] = ws->tmpTs[c][i]; } m->dirL[z + i] = (uint8_t)bestPUMode; }
    HM_SYNC();
    if (pu != numPU - 1) // reconstruction for the next PU, :2632-2660
      par_copy_blk(e->fb.rec[0] + (e->ctuY * 64 + t.y) * ps + e->ctuX * 64 + t.x, ps, ws->reco + t.y * 64 + t.x, 64, n);
  }
  e->mpmZ = -1;
  if (numPU > 1) {
    uint8_t comb[3] = {0, 0, 0};
    for (int p = 0; p < 4; p++) for (int c = 0; c < 3; c++) comb[c] |= (m->cbf[c][cuZ + p * puParts] >> 1) & 1;
    HM_PAR_FOR(o, cuParts) for (int c = 0; c < 3; c++) m->cbf[c][cuZ + o] |= comb[c];
    HM_SYNC();
  }
  cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
  return overallDistY;
}

// ------------------------------------------------------------------------------------------------
// chroma (TEncSearch::xRecurIntraChromaCodingQT :1958-2145, estIntraPredChromaQT :2698-2849)
// ------------------------------------------------------------------------------------------------
HM_DEV inline uint32_t chroma_tu(Shared *e, const TU *t)
{ // the leaf part of xRecurIntraChromaCodingQT
  CtuMeta *m = (&e->meta); const int z = t->cuZ + t->relZ;
  const int fullDepth = t->cuDepth + t->trDepth;
  int checkTS = (t->cW == 4) && (t->log2 == 2);
  if (checkTS) { int nb = 0; for (int s = 0; s < 4; s++) nb += m->ts[0][z + s]; checkTS = nb > 0; }
  const int zc = t->cuZ + t->cRelZ;
  if (t->cW == 4 && !checkTS) {          // both 4x4 blocks at once, one per lane (simt4_chroma_leaf); nothing is coded, so the estimator stays as it is
    cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
    HM_PROF_BEGIN(e, PR_S4CLEAF);
    const uint32_t d2 = HM_UCALL(simt4_chroma_leaf(e, *t));
    HM_PROF_END(e, PR_S4CLEAF);
    return d2;
  }
  uint32_t dist = 0;
  for (int comp = 1; comp < 3; comp++) {
    cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)], &e->cur);
    double singleCost = HM_MAX_DOUBLE, costTmp = 0; uint32_t singleDistC = 0, singleCbfC = 0; int bestTS = 0, bestModeId = 0, currModeId = 0;
    const int total = checkTS ? 2 : 1;
    for (int tsMode = 0; tsMode < total; tsMode++) {
      par_set8(m->ts[comp] + zc, tsMode, t->cParts);
      currModeId++;
      const int isOne = (total == 1), isLast = (currModeId == total);
      const uint32_t distTmp = HM_UCALL(intra_coding_tu_block(e, *t, comp, isOne ? 0 : (tsMode == 0 ? 1 : 2)));
      const uint32_t cbfTmp = (m->cbf[comp][zc] >> t->trDepth) & 1;
      if (tsMode == 1 && cbfTmp == 0) costTmp = HM_MAX_DOUBLE;
      else if (!isOne) { reset_bits(&e->cur); enc_coeff_qt(e, t, comp); costTmp = calc_rd_cost(e, num_bits(&e->cur), distTmp); }   // xGetIntraBitsQTChroma
      if (costTmp < singleCost) {
        singleCost = costTmp; singleDistC = distTmp; bestTS = tsMode; bestModeId = currModeId; singleCbfC = cbfTmp;
        if (!isOne && !isLast) { store_intra_result_qt(e, t, comp); cabac_copy(&e->ws->slot[HM_SLOT(fullDepth, CI_TEMP_BEST)], &e->cur); }
      }
      if (!isOne && !isLast) cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_QT_TRAFO_ROOT)]);
    }
    if (bestModeId < total) {
      load_intra_result_qt(e, t, comp);
      par_set8(m->cbf[comp] + zc, (int)(singleCbfC << t->trDepth), t->cParts);
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(fullDepth, CI_TEMP_BEST)]);
    }
    par_set8(m->ts[comp] + zc, bestTS, t->cParts);
    dist += singleDistC;
  }
  return dist;
}
HM_DEV HM_NOINLINE uint32_t recur_intra_chroma_coding_qt(Shared *e, TU rootv)
{
  HM_ENTRY(e); rootv = hm_uni_struct(rootv);
  const TU *root = &rootv;
  CtuMeta *m = (&e->meta);
  uint32_t dist = 0;
  TuWalk &w = e->walkOuter; walk_begin(&w, root);
  uint8_t (*splitCbf)[2] = e->splitCbf;
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    if (m->tr[z] == t->trDepth) { if (t->cW) dist += chroma_tu(e, t); w.sp--; continue; }
    if (w.next[w.sp] < 0) { w.next[w.sp] = 0; splitCbf[w.sp][0] = splitCbf[w.sp][1] = 0; }
    if (w.next[w.sp] > 0) { // a child just returned
      const TU ch = tu_child(t, w.next[w.sp] - 1, 0);
      for (int c = 1; c < 3; c++) splitCbf[w.sp][c - 1] |= (m->cbf[c][ch.cuZ + ch.relZ] >> ch.trDepth) & 1;
    }
    if (w.next[w.sp] == 4) {
      HM_PAR_FOR(o, t->parts) for (int c = 1; c < 3; c++) if (splitCbf[w.sp][c - 1]) m->cbf[c][z + o] |= (uint8_t)(1 << t->trDepth);
      HM_SYNC();
      w.sp--; continue;
    }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
  }
  return dist;
}
// xSetIntraResultChromaQT, TEncSearch.cpp:2150-2200: visit the chroma leaves
HM_DEV inline void set_intra_result_chroma_qt(Shared *e, const TU *root)
{
  const CtuMeta *m = (&e->meta);
  TuWalk &w = e->walkOuter; walk_begin(&w, root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    if (!t->cW) { w.sp--; continue; }
    if (m->tr[z] == t->trDepth) {
      const int n = t->cW, layer = 5 - t->log2;
      for (int c = 1; c < 3; c++) {
        const int po = HM_PLANE_OFF(c);
        par_copy32(e->cc + po + t->cOff, e->ws->qtCoef[layer] + po + t->cOff, n * n);
        par_copy_blk(e->ws->reco + po + t->cy * 32 + t->cx, 32, e->ws->qtRec[layer] + po + t->cy * 32 + t->cx, 32, n);
      }
      w.sp--; continue;
    }
    if (w.next[w.sp] < 0) w.next[w.sp] = 0;
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 0); w.next[w.sp + 1] = -1; w.sp++;
  }
}
HM_DEV HM_NOINLINE uint32_t est_intra_pred_chroma_qt(Shared *e, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth);
  CtuMeta *m = (&e->meta); WorkSpace *ws = e->ws; const int cuParts = 256 >> (2 * cuDepth);
  const TU t = tu_root(e, cuZ, cuDepth);
  if (cuDepth == 3) {
    // 8x8 CU: one 4x4 block per chroma component; unless that block tries transform skip (which chains Cb into Cr), the five modes x two
    // components are ten independent evaluations: one per lane (hm355_simt4.h)
    const int split = m->tr[cuZ] != 0;
    int tsLuma = 0;
    if (split) for (int s = 0; s < 4; s++) tsLuma += m->ts[0][cuZ + s];
    if (!tsLuma) {
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      HM_PROF_BEGIN(e, PR_S4C);
      const uint32_t d = HM_UCALL(simt4_chroma_cu(e, split ? tu_child(&t, 0, 0) : t, cuZ));
      HM_PROF_END(e, PR_S4C);
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      return d;
    }
  }
  if (cuDepth == 2) {
    // 16x16 CU with an unsplit luma transform: one 8x8 block per chroma component, five modes x two components in lanes (hm355_simt8.h)
    int split = 0;
    for (int i = 0; i < 16; i += 4) split |= m->tr[cuZ + i];
    if (!split) {
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      HM_PROF_BEGIN(e, PR_S8C);
      const uint32_t d = HM_UCALL(simt8_chroma_cu16(e, cuZ));
      HM_PROF_END(e, PR_S8C);
      cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
      return d;
    }
  }
  int bestMode = 0; uint32_t bestDist = 0; double bestCost = HM_MAX_DOUBLE;
  int modeList[5] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX, DM_CHROMA_IDX};       // getAllowedChromaDir, TComDataCU.cpp:1486
  for (int i = 0; i < 4; i++) if (m->dirL[cuZ] == modeList[i]) { modeList[i] = 34; break; }
  for (int mi = 0; mi < 5; mi++) {
    cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
    par_set8(m->dirC + cuZ, modeList[mi], cuParts);
    const uint32_t dist = HM_UCALL(recur_intra_chroma_coding_qt(e, t));
    cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
    const uint32_t bits = HM_UCALL(intra_bits_qt(e, t, 0, 1));
    const double cost = calc_rd_cost(e, bits, dist);
    if (cost < bestCost) {
      bestCost = cost; bestDist = dist; bestMode = modeList[mi];
      set_intra_result_chroma_qt(e, &t);
      HM_PAR_FOR(i, cuParts) for (int c = 1; c < 3; c++) { ws->saveCbf[c][i] = m->cbf[c][cuZ + i]; ws->saveTs[c][i] = m->ts[c][cuZ + i]; }
      HM_SYNC();
    }
  }
  HM_PAR_FOR(i, cuParts) { for (int c = 1; c < 3; c++) { m->cbf[c][cuZ + i] = ws->saveCbf[c][i]; m->ts[c][cuZ + i] = ws->saveTs[c][i]; } m->dirC[cuZ + i] = (uint8_t)bestMode; }
  HM_SYNC();
  cabac_copy(&e->cur, &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
  return bestDist;
}

// ------------------------------------------------------------------------------------------------
// cu_qp_delta (MaxCuDQPDepth 0: the CTU is the quantisation group)
// ------------------------------------------------------------------------------------------------
template <class C> HM_DEV inline void code_delta_qp(Shared *e, C *c, int dqp)
{ // TEncSbac::codeDeltaQP, TEncSbac.cpp:870-895
  const int off = 6 * (e->bitDepth - 8);
  dqp = (dqp + 78 + off + (off / 2)) % (52 + off) - 26 - (off / 2);
  const uint32_t a = (uint32_t)hm_abs(dqp), tu = a < 5 ? a : 5;
  enc_bin(e, c, C_DQP, tu ? 1 : 0);                              // xWriteUnaryMaxSymbol(tu, ctx, 1, CU_DQP_TU_CMAX), :281
  if (tu) { for (uint32_t k = tu; --k;) enc_bin(e, c, C_DQP + 1, 1); if (5 > tu) enc_bin(e, c, C_DQP + 1, 0); }
  if (a >= 5) {                                                  // xWriteEpExGolomb(a - 5, CU_DQP_EG_k = 0), :309
    uint32_t sym = a - 5, count = 0, bins = 0; int nb = 0;
    while (sym >= (1u << count)) { bins = 2 * bins + 1; nb++; sym -= 1u << count; count++; }
    bins = 2 * bins; nb++;
    bins = (bins << count) | sym; nb += (int)count;
    enc_epv(c, bins, nb);
  }
  if (a > 0) enc_epv(c, dqp > 0 ? 0u : 1u, 1);
}
// "dQP: only for CTU once" (TEncEntropy.cpp:343-351): with the first coded block met while TEncCu::m_bEncodeDQP is set
template <class C> HM_DEV inline void code_dqp_if_due(Shared *e, C *c)
{
  if (!e->fb.dqp) return;
  WorkSpace *ws = e->ws;
  if (!HM_UNI(ws->dq.flag)) return;
  code_delta_qp(e, c, HM_UNI(ws->dq.ctuQp) - HM_UNI(ws->dq.refQp));
  HM_SYNC();
  if (hm_lane() == 0) ws->dq.flag = 0;
  HM_SYNC();
}
// TEncCu::xCheckDQP :1742-1763 (RDO_WITHOUT_DQP_BITS 0) on the candidate in e->outBits / outDist / outCost: a candidate of quantisation-group
// size pays for its delta QP when it has a coded block (one without falls back to the predicted QP, which only the QP bookkeeping sees)
HM_DEV inline void check_dqp(Shared *e, int cuZ, int cuDepth)
{
  if (!e->fb.dqp || cuDepth != 0) return;
  const CtuMeta *m = &e->meta;
  if (!(((m->cbf[0][cuZ] | m->cbf[1][cuZ] | m->cbf[2][cuZ]) & 1))) return;
  reset_bits(&e->cur);
  code_delta_qp(e, &e->cur, HM_UNI(e->ws->dq.ctuQp) - HM_UNI(e->ws->dq.refQp));
  e->outBits += num_bits(&e->cur);
  e->outCost = calc_rd_cost(e, e->outBits, e->outDist);
}
// first CU (z order) of the CTU with a coded block, 256 if none (TComDataCU::setQPSubCUs :1763 stops there)
HM_DEV inline int first_coded_cu(const Shared *e)
{
  const CtuMeta *m = &e->meta;
  int best = 256;
  HM_PAR_FOR(z, 256) if (m->cbf[0][z] | m->cbf[1][z] | m->cbf[2][z]) { const int zz = z & ~((256 >> (2 * m->depth[z])) - 1); if (zz < best) best = zz; }
  return -hm_wave_max_i(-best);
}

// ------------------------------------------------------------------------------------------------
// final syntax of a CU (TEncEntropy::xEncodeTransform, TEncEntropy.cpp:222-412)
// ------------------------------------------------------------------------------------------------
template <class C> HM_DEV HM_NOINLINE void encode_cu_syntax(Shared *e, C *c, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); c = hm_uni_ptr(c); HM_ASSUME_LDS(c); // CU-level syntax shared by xCheckRDCostIntra (TEncCu.cpp:1601-1626) and xEncodeCU (:1246-1288), I slice
  const CtuMeta *m = (&e->meta);
  if (e->im) { code_skip_flag(e, c, cuZ); enc_bin(e, c, C_PRED_MODE, 1); }
  if (cuDepth == 3) enc_bin(e, c, C_PART, m->part[cuZ] == SIZE_2Nx2N);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  const unsigned long long profDir0 = __builtin_readcyclecounter();
#endif
  code_intra_dir_luma(e, c, cuZ, 1);
  code_intra_dir_chroma(e, c, cuZ);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  if (EngOf<C>::REAL) { e->prof[PR_SAVE] += __builtin_readcyclecounter() - profDir0; e->profCnt[PR_SAVE] += 1; }
#endif
  const TU root = tu_root(e, cuZ, cuDepth);
  TuWalk &w = e->walkOuter; walk_begin(&w, &root);
  while (w.sp >= 0) {
    TU *t = &w.node[w.sp];
    const int z = t->cuZ + t->relZ;
    const int subdiv = m->tr[z] > t->trDepth;
    if (w.next[w.sp] < 0) {
      if (codes_subdiv_flag(m, t)) enc_bin(e, c, C_SUBDIV + (5 - t->log2), subdiv);
      const int first = t->trDepth == 0;
      for (int comp = 1; comp < 3; comp++)
        if (first || t->cCodeAll)
          if (first || ((m->cbf[comp][z] >> (t->trDepth - 1)) & 1)) code_qt_cbf(e, c, t, comp, subdiv == 0);
      if (!subdiv) {
        code_qt_cbf(e, c, t, 0, 1);
        if (((m->cbf[0][z] | m->cbf[1][z] | m->cbf[2][z]) >> t->trDepth) & 1) code_dqp_if_due(e, c);      // bHaveACodedBlock, TEncEntropy.cpp:343
        for (int comp = 0; comp < 3; comp++) {
          if (comp && !t->cW) continue;
          if (!((m->cbf[comp][z] >> t->trDepth) & 1)) continue;
          const int n = comp ? t->cW : (1 << t->log2);
          const int zc = t->cuZ + (comp ? t->cRelZ : t->relZ);
          const TCoeff *coef = e->cc + HM_PLANE_OFF(comp) + (comp ? t->cOff : z * 16);
          code_coeff_nxn(e, c, coef, n, comp, coef_scan_idx(m, zc, n, comp), m->ts[comp][zc]);
        }
        w.sp--; continue;
      }
      w.next[w.sp] = 0;
    }
    if (w.next[w.sp] == 4) { w.sp--; continue; }
    const int s = w.next[w.sp]++;
    w.node[w.sp + 1] = tu_child(t, s, 1); w.next[w.sp + 1] = -1; w.sp++;
  }
}

// ------------------------------------------------------------------------------------------------
// CU quadtree (TEncCu::xCompressCU :466-1122, xCheckRDCostIntra :1574-1646), explicit stack
// ------------------------------------------------------------------------------------------------
#include "hm355_inter.h"

HM_DEV inline void init_est_data(Shared *e, int cuZ, int cuDepth)
{ // TComDataCU::initEstData, TComDataCU.cpp:484-552
  CtuMeta *m = (&e->meta); const int parts = 256 >> (2 * cuDepth);
  HM_PAR_FOR(i, parts) {
    const int z = cuZ + i;
    m->depth[z] = (uint8_t)cuDepth; m->part[z] = SIZE_NONE; m->pred[z] = MODE_NONE; m->dirL[z] = DC_IDX; m->dirC[z] = 0; m->tr[z] = 0;
    for (int c = 0; c < 3; c++) { m->cbf[c][z] = 0; m->ts[c][z] = 0; }
  }
  HM_PAR_FOR(i, parts * 16) e->cc[cuZ * 16 + i] = 0;
  HM_PAR_FOR(i, parts * 4) { e->cc[4096 + cuZ * 4 + i] = 0; e->cc[5120 + cuZ * 4 + i] = 0; }
  HM_SYNC();
  if (e->im) init_est_data_inter(e, cuZ, cuDepth);
}
HM_DEV inline void meta_copy_range(CtuMeta *d, const CtuMeta *s, int z0, int parts)
{
  HM_PAR_FOR(i, parts) {
    const int z = z0 + i;
    d->depth[z] = s->depth[z]; d->part[z] = s->part[z]; d->pred[z] = s->pred[z]; d->dirL[z] = s->dirL[z]; d->dirC[z] = s->dirC[z]; d->tr[z] = s->tr[z];
    for (int c = 0; c < 3; c++) { d->cbf[c][z] = s->cbf[c][z]; d->ts[c][z] = s->ts[c][z]; }
  }
}
HM_DEV HM_NOINLINE void save_best(Shared *e, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth);
  HM_PROF_BEGIN(e, PR_SAVE);
  Best *b = &e->ws->best[cuDepth]; const int parts = 256 >> (2 * cuDepth);
  meta_copy_range(&b->m, (&e->meta), cuZ, parts);
  if (e->im) imeta_copy_range(&b->im, e->im, cuZ, parts);
  HM_PAR_FOR(i, parts * 16) b->coef[cuZ * 16 + i] = e->cc[cuZ * 16 + i];
  HM_PAR_FOR(i, parts * 4) { b->coef[4096 + cuZ * 4 + i] = e->cc[4096 + cuZ * 4 + i]; b->coef[5120 + cuZ * 4 + i] = e->cc[5120 + cuZ * 4 + i]; }
  const int r = hm_z2r(cuZ), x = (r & 15) * 4, y = (r >> 4) * 4, n = 64 >> cuDepth, l2 = 6 - cuDepth;
  HM_PAR_FOR(i, n * n) { const int yy = i >> l2, xx = i & (n - 1); b->reco[(y + yy) * 64 + x + xx] = e->ws->reco[(y + yy) * 64 + x + xx]; }
  HM_PAR_FOR(i, (n * n) >> 2) {
    const int yy = i >> (l2 - 1), xx = i & ((n >> 1) - 1), o = ((y >> 1) + yy) * 32 + (x >> 1) + xx;
    b->reco[4096 + o] = e->ws->reco[4096 + o]; b->reco[5120 + o] = e->ws->reco[5120 + o];
  }
  HM_SYNC();
  HM_PROF_END(e, PR_SAVE);
}
HM_DEV HM_NOINLINE void restore_best_from(Shared *e, const Best *b, int cuZ, int cuDepth)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); b = hm_uni_ptr(b); HM_ASSUME_GLB(b); // TComDataCU::copyToPic + TEncCu::xCopyYuv2Pic of the unsplit winner
  const int parts = 256 >> (2 * cuDepth);
  meta_copy_range((&e->meta), &b->m, cuZ, parts);
  if (e->im) imeta_copy_range(e->im, &b->im, cuZ, parts);
  HM_PAR_FOR(i, parts * 16) e->cc[cuZ * 16 + i] = b->coef[cuZ * 16 + i];
  HM_PAR_FOR(i, parts * 4) { e->cc[4096 + cuZ * 4 + i] = b->coef[4096 + cuZ * 4 + i]; e->cc[5120 + cuZ * 4 + i] = b->coef[5120 + cuZ * 4 + i]; }
  const int r = hm_z2r(cuZ), x = (r & 15) * 4, y = (r >> 4) * 4, n = 64 >> cuDepth, l2 = 6 - cuDepth;
  Pel *ry = e->fb.rec[0] + (e->ctuY * 64 + y) * e->stride[0] + e->ctuX * 64 + x;
  HM_PAR_FOR(i, n * n) { const int yy = i >> l2, xx = i & (n - 1); ry[yy * e->stride[0] + xx] = b->reco[(y + yy) * 64 + x + xx]; }
  Pel *ru = e->fb.rec[1] + (e->ctuY * 32 + (y >> 1)) * e->stride[1] + e->ctuX * 32 + (x >> 1);
  Pel *rv = e->fb.rec[2] + (e->ctuY * 32 + (y >> 1)) * e->stride[2] + e->ctuX * 32 + (x >> 1);
  HM_PAR_FOR(i, (n * n) >> 2) {
    const int yy = i >> (l2 - 1), xx = i & ((n >> 1) - 1), o = ((y >> 1) + yy) * 32 + (x >> 1) + xx;
    ru[yy * e->stride[1] + xx] = b->reco[4096 + o]; rv[yy * e->stride[2] + xx] = b->reco[5120 + o];
  }
  HM_SYNC();
}
HM_DEV inline void restore_best(Shared *e, int cuZ, int cuDepth) { restore_best_from(e, &e->ws->best[cuDepth], cuZ, cuDepth); }

// xCheckRDCostIntra, TEncCu.cpp:1574-1646; leaves the trial in place
HM_DEV HM_NOINLINE void check_rd_cost_intra(Shared *e, int cuZ, int cuDepth, int partSize)
{
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); partSize = HM_UNI(partSize);
  CtuMeta *m = (&e->meta); const int parts = 256 >> (2 * cuDepth);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  const unsigned long long profCu0 = __builtin_readcyclecounter();
#endif
  init_est_data(e, cuZ, cuDepth);
  HM_PAR_FOR(i, parts) { m->part[cuZ + i] = (uint8_t)partSize; m->pred[cuZ + i] = MODE_INTRA; }
  HM_SYNC();
  HM_PROF_BEGIN(e, PR_LUMA);
  uint32_t d = HM_UCALL(est_intra_pred_qt(e, cuZ, cuDepth));
  HM_PROF_END(e, PR_LUMA);
  { // luma reconstruction of the CU into the picture, TEncCu.cpp:1608
    const int r = hm_z2r(cuZ), x = (r & 15) * 4, y = (r >> 4) * 4, n = 64 >> cuDepth, ps = e->stride[0];
    par_copy_blk(e->fb.rec[0] + (e->ctuY * 64 + y) * ps + e->ctuX * 64 + x, ps, e->ws->reco + y * 64 + x, 64, n);
  }
  { HM_PROF_BEGIN(e, PR_CHROMA); d += HM_UCALL(est_intra_pred_chroma_qt(e, cuZ, cuDepth)); HM_PROF_END(e, PR_CHROMA); }
  reset_bits(&e->cur);
  { HM_PROF_BEGIN(e, PR_ENCCU); encode_cu_syntax(e, &e->cur, cuZ, cuDepth); HM_PROF_END(e, PR_ENCCU); }
  cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)], &e->cur);
  e->outBits = num_bits(&e->cur); e->outDist = d;
  e->outCost = calc_rd_cost(e, e->outBits, e->outDist);
  check_dqp(e, cuZ, cuDepth);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  { const int pid = partSize == SIZE_NxN ? PR_NXN : PR_D0 + cuDepth; e->prof[pid] += __builtin_readcyclecounter() - profCu0; e->profCnt[pid] += 1; }
#endif
}


#include "hm355_inter_cu.h"

// TEncCu::compressCtu -> xCompressCU recursion as a 4-level state machine
HM_DEV HM_NOINLINE void compress_ctu(Shared *e)
{
  HM_ENTRY(e);
  CtuMeta *m = (&e->meta);
  CuFrame *fr = e->cuf; int sp = 0;
  fr[0].cuZ = 0; fr[0].phase = 0; fr[0].parentPart = SIZE_NONE;
  double retCost = 0; uint32_t retBits = 0, retDist = 0;
  while (sp >= 0) {
    CuFrame *f = &fr[sp]; const int cuDepth = sp, cuZ = f->cuZ;
    const int size = 64 >> cuDepth, parts = 256 >> (2 * cuDepth), q = parts >> 2;
    if (f->phase == 0) {
      const int r = hm_z2r(cuZ);
      const int lx = e->ctuX * 64 + (r & 15) * 4, ty = e->ctuY * 64 + (r >> 4) * 4;
      f->boundary = !((lx + size - 1 < e->width) && (ty + size - 1 < e->height));
      f->bestCost = HM_MAX_DOUBLE; f->bestBits = 0; f->bestDist = 0;
      if (!f->boundary && e->im) { // P / B slice: TEncCu.cpp:628-836
        { HM_PROF_BEGIN(e, PR_INTERCU); compress_cu_inter_modes(e, cuZ, cuDepth, sp); HM_PROF_END(e, PR_INTERCU); }
        reset_bits(&e->cur);
        if (cuDepth != 3) enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
        f->bestBits += num_bits(&e->cur);
        f->bestCost = calc_rd_cost(e, f->bestBits, f->bestDist);
      } else if (!f->boundary) {
        check_rd_cost_intra(e, cuZ, cuDepth, SIZE_2Nx2N);
        double c = e->outCost; uint32_t b = e->outBits, d = e->outDist;
        if (c < f->bestCost) { f->bestCost = c; f->bestBits = b; f->bestDist = d; save_best(e, cuZ, cuDepth); cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]); }
        if (cuDepth == 3) {
          check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN);
          c = e->outCost; b = e->outBits; d = e->outDist;
          if (c < f->bestCost) { f->bestCost = c; f->bestBits = b; f->bestDist = d; save_best(e, cuZ, cuDepth); cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]); }
        }
        // split flag of the unsplit candidate, TEncCu.cpp:859-863 (coded on the go-on coder as it stands)
        reset_bits(&e->cur);
        if (cuDepth != 3) enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
        f->bestBits += num_bits(&e->cur);
        f->bestCost = calc_rd_cost(e, f->bestBits, f->bestDist);
      }
      if (cuDepth == 3) { restore_best(e, cuZ, cuDepth); retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue; }
      init_est_data(e, cuZ, cuDepth);
      f->splitBits = 0; f->splitDist = 0; f->sub = 0; f->phase = 1;
    }
    if (f->phase == 1) {
      if (f->sub < 4) {
        const int s = f->sub++;
        const int subZ = cuZ + s * q, r = hm_z2r(subZ);
        const int sx = e->ctuX * 64 + (r & 15) * 4, sy = e->ctuY * 64 + (r >> 4) * 4;
        // TComDataCU::initSubCU, TComDataCU.cpp:555-640
        HM_PAR_FOR(i, q) { m->depth[subZ + i] = (uint8_t)(cuDepth + 1); m->part[subZ + i] = SIZE_NONE; m->pred[subZ + i] = MODE_NONE; }
        HM_SYNC();
        if (sx < e->width && sy < e->height) {
          if (s == 0) cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
          else cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth + 1, CI_NEXT_BEST)]);
          fr[sp + 1].cuZ = (int16_t)subZ; fr[sp + 1].phase = 0;
          // AMP speed-up: the part size of this depth's best mode when it is inter (rpcBestCU->isInter(0), TEncCu.cpp:1026)
          fr[sp + 1].parentPart = (int8_t)((e->im && !f->boundary && e->ws->best[cuDepth].m.pred[cuZ] == MODE_INTER) ? e->ws->best[cuDepth].m.part[cuZ] : SIZE_NONE);
          f->phase = 2; sp++; continue;
        }
        continue;
      }
      if (!f->boundary) {
        reset_bits(&e->cur);
        enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), m->depth[cuZ] > cuDepth);
        f->splitBits += num_bits(&e->cur);
      }
      f->splitCost = calc_rd_cost(e, f->splitBits, f->splitDist);
      if (e->fb.dqp && cuDepth == 0 && first_coded_cu(e) < 256) {   // the split candidate of quantisation-group size pays for its delta QP, TEncCu.cpp:1052-1085
        reset_bits(&e->cur);
        code_delta_qp(e, &e->cur, HM_UNI(e->ws->dq.ctuQp) - HM_UNI(e->ws->dq.refQp));
        f->splitBits += num_bits(&e->cur);
        f->splitCost = calc_rd_cost(e, f->splitBits, f->splitDist);
      }
      cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)], &e->ws->slot[HM_SLOT(cuDepth + 1, CI_NEXT_BEST)]);
      if (f->splitCost < f->bestCost) {
        f->bestCost = f->splitCost; f->bestBits = f->splitBits; f->bestDist = f->splitDist;
        cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
      } else restore_best(e, cuZ, cuDepth);
      retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue;
    }
    if (f->phase == 2) { // a sub-CU returned
      f->splitBits += retBits; f->splitDist += retDist; f->phase = 1;
      // The split candidate's cost only grows from here (bits and distortion add up, calcRdCost is monotone in both) and it has to come in
      // strictly below the unsplit best (TEncCu.cpp:1704): once the sub-CUs so far are no cheaper, the rest cannot change the decision.
      // One thing of the skipped work does reach outside: the split flag of the parent's split candidate is priced on the estimator "as it
      // stands" (TEncCu.cpp:1042-1047), i.e. as this CU's last sub-CU left it - so a CU only stops early when a later sibling (which reloads
      // the estimator from its slot) follows it, or at the CTU root.
      // I slices only: in P / B slices the integer vector of the last 2Nx2N motion search seeds the next CU's search whatever CU that was
      // (m_integerMv2Nx2N, TEncSearch.cpp:3880-3888), so a skipped sub-CU would change its successors.
      bool laterSibling = sp == 0;
      if (sp > 0) {
        const int pq = parts, pz = fr[sp - 1].cuZ;
        for (int s2 = fr[sp - 1].sub; s2 < 4; s2++) {
          const int r2 = hm_z2r(pz + s2 * pq);
          laterSibling |= (e->ctuX * 64 + (r2 & 15) * 4 < e->width) && (e->ctuY * 64 + (r2 >> 4) * 4 < e->height);
        }
      }
      if (f->sub < 4 && !e->im && laterSibling && !f->boundary && !(calc_rd_cost(e, f->splitBits, f->splitDist) < f->bestCost)) {
        restore_best(e, cuZ, cuDepth);
        retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--;
      }
      continue;
    }
  }
  e->outCost = retCost; e->outBits = retBits; e->outDist = retDist;
}

// TEncCu::xEncodeCU, TEncCu.cpp:1185-1295: re-encode the decided CTU to advance the contexts
template <class C> HM_DEV HM_NOINLINE void encode_ctu(Shared *e, C *c, int lastCtuOfSlice)
{
  HM_ENTRY(e); lastCtuOfSlice = HM_UNI(lastCtuOfSlice); c = hm_uni_ptr(c); HM_ASSUME_LDS(c);
  const CtuMeta *m = (&e->meta);
  int16_t stackZ[4]; int8_t stackNext[4]; int sp = 0;
  stackZ[0] = 0; stackNext[0] = -1;
  while (sp >= 0) {
    const int depth = sp, z = stackZ[sp], size = 64 >> depth;
    const int r = hm_z2r(z);
    const int lx = e->ctuX * 64 + (r & 15) * 4, ty = e->ctuY * 64 + (r >> 4) * 4;
    const int inside = (lx + size - 1 < e->width) && (ty + size - 1 < e->height);
    if (stackNext[sp] < 0) {
      if (inside && depth != 3) enc_bin(e, c, C_SPLIT + ctx_split_flag(e, z, depth), m->depth[z] > depth);
      if (!((depth < m->depth[z] && depth < 3) || !inside)) {
        if (m->pred[z] == MODE_INTER) encode_cu_syntax_inter(e, c, z, depth, 1); else encode_cu_syntax(e, c, z, depth);
        // finishCU, TEncCu.cpp:1130-1147
        const int lastX = ((lx + size) % 64 == 0) || (lx + size == e->width), lastY = ((ty + size) % 64 == 0) || (ty + size == e->height);
        if (lastX && lastY && !lastCtuOfSlice) enc_trm(e, c, 0);
        sp--; continue;
      }
      stackNext[sp] = 0;
    }
    if (stackNext[sp] == 4) { sp--; continue; }
    const int s = stackNext[sp]++;
    const int q = (256 >> (2 * depth)) >> 2, sz = z + s * q, rr = hm_z2r(sz);
    const int sx = e->ctuX * 64 + (rr & 15) * 4, sy = e->ctuY * 64 + (rr >> 4) * 4;
    if (sx < e->width && sy < e->height) { stackZ[sp + 1] = (int16_t)sz; stackNext[sp + 1] = -1; sp++; }
  }
}

// ------------------------------------------------------------------------------------------------
// one CTU of TEncSlice::compressSlice (TEncSlice.cpp:724-892)
// ------------------------------------------------------------------------------------------------
#if !defined(HM355_HOSTSIM)
#include "hm355_team.h"
#endif
HM_DEV inline void process_ctu(Shared *e, const Params *P, const WorkItem *it, int wsIndex, Team *team = 0)
{
  // uniform context (every lane writes the same values)
  e->P = P; e->fb = P->frames[it->frame]; e->ws = P->ws + wsIndex; e->tab = P->tab;
  e->width = P->width; e->height = P->height; e->bitDepth = P->bitDepth; e->wCtu = P->wCtu; e->mpmZ = -1; e->s8Reuse = 0;
  for (int c = 0; c < 3; c++) e->stride[c] = P->stride[c];
  e->ctuX = it->ctuX; e->ctuY = it->ctuY; e->ctuAddr = it->ctuY * P->wCtu + it->ctuX;
  e->cc = e->fb.coef + (size_t)e->ctuAddr * HM_COEF_CTU;
  e->im = e->fb.imeta ? e->fb.imeta + e->ctuAddr : (InterMeta *)0;
  HM_SYNC();
  if (e->im) {
    // TEncSearch::m_integerMv2Nx2N persists from CTU to CTU in coding order.  A CTU whose 64x64 CU lies inside the picture
    // overwrites every entry (2Nx2N search at depth 0, all reference indices) before it reads one, so only picture-boundary
    // CTUs take the state of their predecessor -- which is what lets inter slices run as a WPP wavefront.
    const int bnd = it->ctuX * 64 + 63 >= P->width || it->ctuY * 64 + 63 >= P->height;
    const MvD *src = e->ctuAddr == 0 ? e->fb.ip->integerMv2Nx2N[0] : e->fb.intMv + (size_t)(e->ctuAddr - 1) * 32;
    HM_PAR_FOR(i, 32) { MvD v; v.x = v.y = 0; if (bnd) v = src[i]; e->ws->intMv[i >> 4][i & 15] = v; }
    HM_SYNC();
  }
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  for (int i = 0; i < HM_PROF_N; i++) { e->prof[i] = 0; e->profCnt[i] = 0; }
#endif
  HM_PROF_BEGIN(e, PR_TOTAL);
#if defined(HM355_HOSTSIM)
  load_tables();                       // (the kernels load them once per workgroup)
#endif
  const int a = e->ctuAddr, numCtus = P->wCtu * P->hCtu;
  { // TComDataCU::initCtu, TComDataCU.cpp:357-470
    CtuMeta *m = (&e->meta);
    HM_PAR_FOR(z, 256) {
      m->depth[z] = 0; m->part[z] = SIZE_NONE; m->pred[z] = MODE_NONE; m->dirL[z] = DC_IDX; m->dirC[z] = 0; m->tr[z] = 0;
      for (int c = 0; c < 3; c++) { m->cbf[c][z] = 0; m->ts[c][z] = 0; }
    }
    HM_SYNC();
    if (e->im) init_est_data_inter(e, 0, 0);
  }
  const int sliceQp = e->fb.qp;
  if (e->fb.dqp) {
    // cu_qp_delta: the QP of this CTU (TEncCu::xComputeQP :1154 / the rate control's, TEncSlice.cpp:767-808) replaces the slice's in the quantiser
    // parameters; its predictor is the QP of the last CU coded before it in this CTU row / slice (both neighbouring quantisation groups lie
    // outside the CTU: TComDataCU::getRefQP :1413, getLastCodedQP :1434-1468); TEncCu::m_bEncodeDQP arrives from the previous CTU in coding order
    const DqpPic *dp = e->fb.dqp;
    const int rowStart = e->ctuX == 0 && P->wpp;
    const int q = dp->ctuQp[a];
    const int refQp = (a == 0 || rowStart) ? sliceQp : dp->out[a - 1].lastQp;
    const int flag = a == 0 ? dp->flagIn : (rowStart ? dp->rowFlag[e->ctuY] : dp->out[a - 1].flagOut);
    const QpTab *t = &dp->tab[q + 12];
    e->fb.qp = q;
    for (int k = 0; k < 2; k++) { e->fb.qpPer[k] = t->qpPer[k]; e->fb.qpRem[k] = t->qpRem[k]; e->fb.rdFactor[k] = t->rdFactor[k]; for (int l = 0; l < 4; l++) e->fb.errScale[k][l] = t->errScale[k][l]; }
    if (hm_lane() == 0) { e->ws->dq.ctuQp = q; e->ws->dq.refQp = refQp; e->ws->dq.flag = flag; }
    HM_SYNC();
  }
  // CABAC state hand-off (TEncSlice.cpp:733-761)
  Cabac *cb0 = &e->ws->slot[HM_SLOT(0, CI_CURR_BEST)];
  const int initType = e->im ? e->fb.ip->cabacInitType : 2;   // context table of the slice type (TEncSbac::resetEntropy :106-115)
  if (a == 0) cabac_init(cb0, sliceQp, initType);
  else if (e->ctuX == 0 && P->wpp) {
    cabac_init(cb0, sliceQp, initType);
    if (e->ctuY > 0 && P->wCtu > 1) { // contexts of the 2nd CTU of the row above, fresh bit accumulator
      const Cabac *src = e->fb.endState + ((e->ctuY - 1) * P->wCtu + 1);
      HM_PAR_FOR(i, HM_NUM_CTX) cb0->s[i] = src->s[i];
      HM_SYNC();
    }
  } else cabac_copy(cb0, e->fb.endState + (a - 1));
  cabac_copy(&e->cur, cb0);
#if !defined(HM355_HOSTSIM)
  // (a team hands the 64x64 candidate to a helper while it goes on: with m_bEncodeDQP set the later candidates would need to know whether that
  // one consumed it, so such a CTU -- rare -- is searched by the main wavefront alone)
  if (team && e->im && !e->fb.dqp) compress_ctu_team_inter(e);
  else if (team && !e->im && !(e->fb.dqp && HM_UNI(e->ws->dq.flag))) compress_ctu_team(e); else
#endif
  compress_ctu(e);
  e->fb.stat[a].cost = e->outCost; e->fb.stat[a].bits = e->outBits; e->fb.stat[a].dist = e->outDist;
  // TEncCu::encodeCtu on m_pppcRDSbacCoder[0][CI_CURR_BEST] (the search never writes that snapshot), TEncSlice.cpp:818-825:
  // run it on the LDS-resident coder and hand the end state to the next CTU
  cabac_copy(&e->cur, cb0);
  reset_bits(&e->cur);
  if (e->fb.dqp) { if (hm_lane() == 0) e->ws->dq.flag = 1; HM_SYNC(); }      // TEncCu::encodeCtu :358-361
  encode_ctu(e, &e->cur, a == numCtus - 1);
  cabac_copy(e->fb.endState + a, &e->cur);
  if (e->fb.dqp) {
    const int firstZ = first_coded_cu(e), q = HM_UNI(e->ws->dq.ctuQp), refQp = HM_UNI(e->ws->dq.refQp), flag = HM_UNI(e->ws->dq.flag);
    if (hm_lane() == 0) { CtuDqp o; o.qp = (int8_t)q; o.refQp = (int8_t)refQp; o.lastQp = (int8_t)(firstZ < 256 ? q : refQp); o.flagOut = (uint8_t)flag; o.firstZ = (int16_t)firstZ; o.pad = 0; e->fb.dqp->out[a] = o; }
    HM_SYNC();
  }
  if (e->im) { HM_PAR_FOR(i, 32) e->fb.intMv[(size_t)a * 32 + i] = e->ws->intMv[i >> 4][i & 15]; HM_SYNC(); }   // carried to the next CTU in coding order
  { // decision arrays back to HBM (TComDataCU::copyToPic of the whole CTU)
    const uint32_t *src = (const uint32_t *)&e->meta; uint32_t *dst = (uint32_t *)(e->fb.meta + a);
    HM_PAR_FOR(i, (int)(sizeof(CtuMeta) / 4)) dst[i] = src[i];
    HM_SYNC();
  }
  HM_PROF_END(e, PR_TOTAL);
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  if (hm_lane() == 0 && P->prof) for (int i = 0; i < HM_PROF_N; i++) { atomicAdd(P->prof + i, e->prof[i]); atomicAdd(P->prof + HM_PROF_N + i, e->profCnt[i]); }
#endif
}
