// hm355 -- gfx950 kernels and the C ABI of include/hm355.h.
//
// One workgroup (one 64-lane wavefront) searches one CTU; a launch carries every CTU of every picture
// in the batch whose dependencies are complete (see hm355_build_schedule).  Launches of consecutive
// steps are ordered by the stream, which is also what makes the neighbours' reconstruction, decision
// arrays and CABAC hand-off visible (kernel boundary == device-scope release/acquire).
//
// There is no CPU path in this library: every entry point needs a working HIP device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "hm355_core.h"
#include "hm355_dbk.h"
#include "hm355_sao.h"
#include "hm355_bits_kernel.h"
#include "hm355_ingest.h"
#include "hm355_host_common.h"
#include "../../include/hm355.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// Persistent CTU scheduler.  The work list holds every CTU of every picture of the batch in a dependency-
// respecting (topological) order; a workgroup (one wavefront) repeatedly takes the next ticket, waits until the
// CTUs it depends on have published their results, searches its CTU and publishes.  Because tickets are taken
// in topological order and a workgroup only takes a ticket while it is running, the oldest unfinished ticket
// always has its dependencies satisfied: the grid drains for any grid size and any placement.
//   dependencies of CTU (x,y):  left (x-1,y);  above-right (x+1,y-1) (above at the right picture edge)  [WPP]
//                               previous CTU in raster order (CABAC chain)                               [no WPP]
// Hand-off between workgroups follows the agent-scope release/acquire recipe: all stores of the wave, release
// fence, s_waitcnt, relaxed flag store; consumer: relaxed poll by one lane, acquire fence, plain loads.
#define HM_SPIN_TIMEOUT_TICKS (40ull * 100000000ull)   /* 40 s of the 100 MHz wall clock without ANY CTU of the launch being published: bounds every spin (the oldest unfinished ticket always runs, and one CTU search takes at most a second or two) */
typedef __attribute__((address_space(1))) unsigned int gu32;   // global address space: never a flat access

// lane 0 polls the flag of one dependency (relaxed, agent scope); returns non-zero when the run must be abandoned
__device__ __attribute__((noinline)) int hm355_wait_flag(const unsigned int *flag, unsigned int *abortWord, unsigned int epoch)
{
  int bad = 0;
  if (hm_lane() == 0) {
    const gu32 *f = (const gu32 *)flag; gu32 *ab = (gu32 *)abortWord;
    unsigned long long t0 = wall_clock64();
    unsigned int seen = __hip_atomic_load(ab + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // CTUs the launch has published so far
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(32);
      int stop = __hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      if (!stop && wall_clock64() - t0 > HM_SPIN_TIMEOUT_TICKS) {
        // a serial CABAC chain (no WPP) legitimately keeps the last ticket holders waiting for minutes: give up only when the
        // whole launch has published nothing for the length of the timeout
        const unsigned int now = __hip_atomic_load(ab + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (now != seen) { seen = now; t0 = wall_clock64(); } else stop = 1;
      }
      if (stop) {
        __hip_atomic_store(ab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tell everybody to drain
        bad = 1; break;
      }
    }
  }
  return __shfl(bad, 0, 64);
}

// A workgroup holds HM_CTU_WAVES wavefronts, each an independent CTU search with its own ticket loop, LDS state and HBM workspace; they share
// the read-only tables (LdsTables) and nothing else -- no workgroup barrier after the tables are loaded.  Two such workgroups fit a CU: 12 searches.
extern "C" __global__ void __launch_bounds__(64 * HM_CTU_WAVES, 3) hm355_ctu_kernel(const Params *P, const WorkItem *items, int total, unsigned int *sched, unsigned int epoch)
{
  __shared__ WorkItem curItems[HM_CTU_WAVES];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave == 0) load_tables();
  __syncthreads();
  WorkItem &curItem = curItems[wave];
  for (;;) {
    int idx = 0;
    if (hm_lane() == 0) idx = (int)atomicAdd(&sched[0], 1u);
    idx = __shfl(idx, 0, 64);
    if (idx >= total) break;
    if (hm_lane() == 0) curItem = items[idx];
    HM_SYNC();
    const int cx = curItem.ctuX, cy = curItem.ctuY, wCtu = P->wCtu;
    const unsigned int *done = P->frames[curItem.frame].done;
    const int a = cy * wCtu + cx;
    int dep0 = -1, dep1 = -1;
    if (P->wpp) {
      if (cx > 0) dep0 = a - 1;
      if (cy > 0) dep1 = (cy - 1) * wCtu + (cx + 1 < wCtu ? cx + 1 : cx);
    } else if (a > 0) dep0 = a - 1;
    // inter slice: a picture-boundary CTU takes the 2Nx2N integer-MV state of its predecessor in coding order (process_ctu)
    const int dep2 = (P->wpp && cx == 0 && cy > 0 && P->frames[curItem.frame].imeta && (63 >= P->width || cy * 64 + 63 >= P->height)) ? a - 1 : -1;
    int bad = 0;
    if (dep0 >= 0) bad = hm355_wait_flag(done + dep0, sched + 1, epoch);
    if (!bad && dep1 >= 0) bad = hm355_wait_flag(done + dep1, sched + 1, epoch);
    if (!bad && dep2 >= 0) bad = hm355_wait_flag(done + dep2, sched + 1, epoch);
    if (bad) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    process_ctu(&g_shs[wave], P, &curItem, (int)blockIdx.x * HM_CTU_WAVES + wave);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (hm_lane() == 0) { __hip_atomic_store((gu32 *)(done + a), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add((gu32 *)(sched + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    HM_SYNC();
  }
}

// The same scheduler with a TEAM of wavefronts per CTU (hm355_team.h): wave 0 takes the tickets, waits for the dependencies and runs the
// reference's recursion; waves 1.. evaluate the unsplit candidates it hands them.  Used for launches that cannot fill the device with
// one-wavefront searches (a few pictures, or pictures whose CABAC state chains through every CTU), where the time of ONE CTU search is
// what the launch takes.
extern "C" __global__ void __launch_bounds__(64 * HM_TEAM, 3) hm355_ctu_team_kernel(const Params *P, const WorkItem *items, int total, unsigned int *sched, unsigned int epoch)
{ // blockDim.x = 64 * HM_TEAM_I (I slices only) or 64 * HM_TEAM; dynamic LDS = HM_TEAM_LDS_BYTES(waves)
  Team *T = HM_TEAM_PTR();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < HM_TEAM - 1) { T->box[threadIdx.x].reqSeq = 0; T->box[threadIdx.x].doneSeq = 0; }
  if (threadIdx.x == 0) { T->quit = 0; T->dead = 0; T->abortWord = sched + 1; T->waves = (int)(blockDim.x >> 6); }
  if (wave == 0) load_tables();
  __syncthreads();
  if (wave != 0) {
    team_helper(T, wave, wave <= HM_TEAM_HELPERS ? P->teamWin + ((size_t)blockIdx.x * HM_TEAM_HELPERS + (size_t)(wave - 1)) * P->teamWinStride : (Pel *)0);
    return;
  }
  for (;;) {
    int idx = 0;
    if (threadIdx.x == 0) idx = (int)atomicAdd(&sched[0], 1u);
    idx = __shfl(idx, 0, 64);
    if (idx >= total) break;
    if (threadIdx.x == 0) T->item = items[idx];
    HM_SYNC();
    const int cx = T->item.ctuX, cy = T->item.ctuY, wCtu = P->wCtu;
    const unsigned int *done = P->frames[T->item.frame].done;
    const int a = cy * wCtu + cx;
    int dep0 = -1, dep1 = -1;
    if (P->wpp) {
      if (cx > 0) dep0 = a - 1;
      if (cy > 0) dep1 = (cy - 1) * wCtu + (cx + 1 < wCtu ? cx + 1 : cx);
    } else if (a > 0) dep0 = a - 1;
    const int dep2 = (P->wpp && cx == 0 && cy > 0 && P->frames[T->item.frame].imeta && (63 >= P->width || cy * 64 + 63 >= P->height)) ? a - 1 : -1;   // as in hm355_ctu_kernel
    int bad = 0;
    if (dep0 >= 0) bad = hm355_wait_flag(done + dep0, sched + 1, epoch);
    if (!bad && dep1 >= 0) bad = hm355_wait_flag(done + dep1, sched + 1, epoch);
    if (!bad && dep2 >= 0) bad = hm355_wait_flag(done + dep2, sched + 1, epoch);
    if (bad) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    process_ctu(&T->sh[0], P, &T->item, (int)blockIdx.x * T->waves, T);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) { __hip_atomic_store((gu32 *)(done + a), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add((gu32 *)(sched + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    HM_SYNC();
    if (team_ld(&T->dead)) break;
  }
  HM_TEAM_RELEASE();
  team_st(&T->quit, 1u);
}

// batched distortion primitives: one wavefront per n x n block pair
extern "C" __global__ void __launch_bounds__(64) hm355_dist_kernel(int kind, int n, int bitDepth, int count, const Pel *org, const Pel *cur, uint32_t *out)
{
  for (int b = (int)blockIdx.x; b < count; b += (int)gridDim.x) {
    const Pel *o = org + (size_t)b * n * n, *c = cur + (size_t)b * n * n;
    uint32_t v;
    if (kind == 0) v = dist_sad(o, n, c, n, n, 0, bitDepth);
    else if (kind == 3) v = dist_sad(o, n, c, n, n, 1, bitDepth);
    else if (kind == 1) v = dist_sse(o, n, c, n, n, bitDepth);
    else v = dist_hads(o, n, c, n, n, bitDepth);
    if (hm_lane() == 0) out[b] = v;
  }
}
// batched transforms: LDS-staged, one wavefront per block
extern "C" __global__ void __launch_bounds__(64) hm355_transform_kernel(int inverse, int n, int bitDepth, int useDst, int count, const int32_t *in, int32_t *out)
{
  Shared &sh = g_sh;
  load_tables();
  const int l2 = hm_log2(n);
  for (int b = (int)blockIdx.x; b < count; b += (int)gridDim.x) {
    const int32_t *src = in + (size_t)b * n * n; int32_t *dst = out + (size_t)b * n * n;
    HM_PAR_FOR(i, n * n) sh.bufA[(i >> l2) * HM_TSTRIDE + (i & (n - 1))] = src[i];
    HM_SYNC();
    if (inverse) inv_transform(&sh, n, useDst, bitDepth); else fwd_transform(&sh, n, useDst, bitDepth);
    HM_PAR_FOR(i, n * n) dst[i] = sh.bufA[(i >> l2) * HM_TSTRIDE + (i & (n - 1))];
    HM_SYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct hm355_ref {       // a finished picture as later pictures reference it (device buffers owned by this object)
  RefPicDev dev;
  std::vector<void *> owned;
};
struct Slot {           // one picture resident in HBM
  FrameBuf fb;          // device pointers + slice parameters (host copy)
  InterMeta *imeta;     // motion arrays of the slot (allocated on first inter use; kept for the deblocking pass)
  Pel *saoSrc[3]; SaoStat *saoStat; SaoCand *saoCand; SaoBlk *saoCoded, *saoRecon;   // SAO working buffers (allocated on first use)
  uint8_t *rawIn, *rawOut;   // file frames as they are on disk (ingest / output, allocated on first use)
  uint8_t *bitsRaw, *bitsPacked; uint32_t *bitsSizes; CabacW *bitsSync; uint32_t *bitsFlag; InterPic *bitsIp;   // bitstream pass (allocated on first use)
  // cu_qp_delta (hm355_set_dqp): device state of the picture, allocated on first use; dqpOn: the next searches of the slot run with it
  DqpPic *dDqp; int8_t *dCtuQp; CtuDqp *dDqpOut; uint8_t *dRowFlag; int dqpOn, dqpFlagIn;
  std::vector<int8_t> ctuQp; std::vector<uint8_t> rowFlag; hm355_slice_desc lastSlice;
};
#define HM_BITS_CAP_PER_CTU 16384u   /* bytes reserved per CTU in the raw substream buffers: above the raw size of a 10-bit 4:2:0 CTU (7.7 KB) */
#define HM_MAX_LANES 4
struct Lane {           // one launch of the search in flight: its own stream, scratch areas, work list and scheduler words
  hipStream_t stream; hipEvent_t ev0, ev1;
  Params *dP;           // device copy of the kernel parameters with this lane's scratch areas
  WorkSpace *dWs; size_t wsCount;
  WorkItem *dItems; size_t itemsCap;
  unsigned int *dSched; // [0] ticket, [1] abort, [2] CTUs published by the launch (the spin timeout watches it)
  std::vector<WorkItem> items; std::vector<int> stepStart; std::vector<FrameBuf> fbs;
  long long key[5]; int keyValid, fewWaves;
  int busy, grid, inFixup;
  Pel *dTeamWin; size_t teamCap;   // team launches (hm355_team.h): the helpers' reconstruction windows, for teamCap teams
};
struct hm355_ctx {
  hm355_seq_cfg cfg;
  Params hp;            // host copy of the kernel parameters
  Params *dP;
  Tables *dTab;
  FrameBuf *dFrames;
  WorkSpace *dWs; size_t wsCount;
  uint8_t *arena;       // the pictures' planes, decision arrays, coefficients, statistics, CABAC states, done words: one allocation
  int teamLdsSet;
  unsigned char *hStage; size_t hStageBytes;   // pinned host staging of hm355_download (one picture's results), allocated on first use
  unsigned int *dSched; unsigned int epoch;   // dSched: [0] ticket, [1] abort, [2] published CTUs of the search launch; [8] ticket, [9] abort of the bitstream launch
  std::vector<Slot> slots;
  Lane lane[HM_MAX_LANES];   // lane 0 is the context's own stream / scratch (every blocking entry point); 1.. are created on first use
  hipStream_t stream; hipEvent_t ev0, ev1;
  double lastKernelMs; int lastLaunches; int laneShare;
  std::string err;
  int numCtus;
  void *staging; size_t stagingBytes;
  DbkParams *dDbk;      // [max_batch] deblocking parameters of the pictures in the slots
  SaoParams *dSao;      // [max_batch] SAO parameters / results of the pictures in the slots
  BitsParams *dBits;    // [max_batch] bitstream pass parameters / results
  IngestParams *dIngest; // [max_batch] ingest / output parameters
};

#define HM_CHECK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return HM355_ERR_DEVICE; } } while (0)

static int fail(hm355_ctx *c, int code, const char *msg) { if (c) c->err = msg; return code; }

#ifndef HM355_BUILD_ID
#define HM355_BUILD_ID "unknown"
#endif
extern "C" const char *hm355_build_id(void) { return HM355_BUILD_ID; }
extern "C" const char *hm355_last_error(const hm355_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

static int maxItemsPerStep(int wCtu, int hCtu, int wpp, int frames)
{
  if (!wpp) return frames;
  int best = 0;
  for (int s = 0; s < wCtu + 2 * (hCtu - 1); s++) { int c = 0; for (int y = 0; y < hCtu; y++) { int x = s - 2 * y; if (x >= 0 && x < wCtu) c++; } if (c > best) best = c; }
  return best * frames;
}

extern "C" int hm355_create(const hm355_seq_cfg *cfg, hm355_ctx **out)
{
  if (!cfg || !out) return HM355_ERR_ARG;
  *out = NULL;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || (cfg->bit_depth != 8 && cfg->bit_depth != 10) ||
      cfg->ctu_size != 64 || cfg->max_cu_depth != 4 || cfg->tu_log2_max != 5 || cfg->tu_log2_min != 2 || cfg->tu_max_depth_intra != 3 ||
      cfg->max_batch < 1 || (cfg->wavefront_synchro != 0 && cfg->wavefront_synchro != 1))
    return HM355_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HM355_ERR_NO_DEVICE;
  hm355_ctx *c = new hm355_ctx();
  c->cfg = *cfg; c->laneShare = 1; c->lastKernelMs = 0; c->lastLaunches = 0; c->staging = NULL; c->stagingBytes = 0; c->dDbk = NULL; c->dSao = NULL; c->dBits = NULL; c->dIngest = NULL;
  c->arena = NULL; c->dP = NULL; c->dTab = NULL; c->dFrames = NULL; c->dWs = NULL; c->wsCount = 0; c->dSched = NULL; c->epoch = 0; c->teamLdsSet = 0; c->hStage = NULL; c->hStageBytes = 0;
  for (int l = 0; l < HM_MAX_LANES; l++) { Lane &L = c->lane[l]; L.stream = NULL; L.ev0 = L.ev1 = NULL; L.dP = NULL; L.dWs = NULL; L.wsCount = 0; L.dItems = NULL; L.itemsCap = 0; L.dSched = NULL; L.keyValid = 0; L.fewWaves = -1; L.busy = 0; L.grid = 0; L.inFixup = 0; L.dTeamWin = NULL; L.teamCap = 0; }
  Params &P = c->hp; memset(&P, 0, sizeof(P));
  P.width = cfg->width; P.height = cfg->height; P.bitDepth = cfg->bit_depth; P.wpp = cfg->wavefront_synchro;
  P.wCtu = (cfg->width + 63) / 64; P.hCtu = (cfg->height + 63) / 64;
  P.stride[0] = P.wCtu * 64; P.stride[1] = P.stride[2] = P.wCtu * 32;
  c->numCtus = P.wCtu * P.hCtu;
  *out = c;   // from here on the caller destroys on failure
  HM_CHECK(c, hipStreamCreate(&c->stream));
  HM_CHECK(c, hipEventCreate(&c->ev0)); HM_CHECK(c, hipEventCreate(&c->ev1));
  Tables *ht = new Tables; hm355_build_tables(ht);
  hipError_t e = hipMalloc((void **)&c->dTab, sizeof(Tables));
  if (e == hipSuccess) e = hipMemcpy(c->dTab, ht, sizeof(Tables), hipMemcpyHostToDevice);
  delete ht;
  HM_CHECK(c, e);
  // one scratch area per resident workgroup of the persistent grid: 256 CUs x 8 single-wave workgroups, or fewer when the batch is small
  c->wsCount = (size_t)c->numCtus * (size_t)cfg->max_batch * HM_TEAM; if (c->wsCount > 3072) c->wsCount = 3072;   // 12 searches per CU x 256 CUs is the most that can be resident; a small batch runs as teams of HM_TEAM wavefronts per CTU
  HM_CHECK(c, hipMalloc((void **)&c->dSched, 64)); HM_CHECK(c, hipMemset(c->dSched, 0, 64));
  HM_CHECK(c, hipMalloc((void **)&c->dWs, c->wsCount * sizeof(WorkSpace)));
  HM_CHECK(c, hipMalloc((void **)&c->dFrames, sizeof(FrameBuf) * cfg->max_batch));
  HM_CHECK(c, hipMalloc((void **)&c->dP, sizeof(Params)));
  c->slots.resize(cfg->max_batch);
  { // the pictures' buffers come out of ONE allocation (a batch of thousands of small pictures used to mean tens of thousands of hipMallocs)
    size_t planeBytes[3], off = 0;
    for (int k = 0; k < 3; k++) planeBytes[k] = ((size_t)P.stride[k] * P.hCtu * (k ? 32 : 64) * sizeof(Pel) + 255) & ~(size_t)255;
    const size_t metaBytes = (sizeof(CtuMeta) * c->numCtus + 255) & ~(size_t)255, coefBytes = (sizeof(TCoeff) * (size_t)c->numCtus * HM_COEF_CTU + 255) & ~(size_t)255;
    const size_t statBytes = (sizeof(CtuStat) * c->numCtus + 255) & ~(size_t)255, endBytes = (sizeof(Cabac) * c->numCtus + 255) & ~(size_t)255;
    const size_t doneBytes = (sizeof(uint32_t) * c->numCtus + 255) & ~(size_t)255;
    const size_t slotBytes = 2 * (planeBytes[0] + planeBytes[1] + planeBytes[2]) + metaBytes + coefBytes + statBytes + endBytes + doneBytes;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && slotBytes * (size_t)cfg->max_batch > freeB) {
      c->err = "hm355_create: max_batch pictures of this size do not fit the device memory"; return HM355_ERR_NOMEM; }
    HM_CHECK(c, hipMalloc((void **)&c->arena, slotBytes * (size_t)cfg->max_batch));
    HM_CHECK(c, hipMemset(c->arena, 0, slotBytes * (size_t)cfg->max_batch));      // planes (padding included) and the done words start at zero
    for (int s = 0; s < cfg->max_batch; s++) {
      FrameBuf &fb = c->slots[s].fb; memset(&fb, 0, sizeof(fb));
      c->slots[s].imeta = NULL; c->slots[s].saoSrc[0] = c->slots[s].saoSrc[1] = c->slots[s].saoSrc[2] = NULL; c->slots[s].saoStat = NULL; c->slots[s].saoCand = NULL; c->slots[s].saoCoded = c->slots[s].saoRecon = NULL;
      c->slots[s].rawIn = c->slots[s].rawOut = NULL;
      c->slots[s].bitsRaw = c->slots[s].bitsPacked = NULL; c->slots[s].bitsSizes = NULL; c->slots[s].bitsSync = NULL; c->slots[s].bitsFlag = NULL; c->slots[s].bitsIp = NULL;
      c->slots[s].dDqp = NULL; c->slots[s].dCtuQp = NULL; c->slots[s].dDqpOut = NULL; c->slots[s].dRowFlag = NULL; c->slots[s].dqpOn = 0; c->slots[s].dqpFlagIn = 0;
      uint8_t *p = c->arena + off;
      for (int k = 0; k < 3; k++) { fb.org[k] = (Pel *)p; p += planeBytes[k]; fb.rec[k] = (Pel *)p; p += planeBytes[k]; }
      fb.meta = (CtuMeta *)p; p += metaBytes; fb.coef = (TCoeff *)p; p += coefBytes; fb.stat = (CtuStat *)p; p += statBytes;
      fb.endState = (Cabac *)p; p += endBytes; fb.done = (uint32_t *)p; p += doneBytes;
      off += slotBytes;
    }
  }
  P.tab = c->dTab; P.ws = c->dWs; P.frames = c->dFrames; P.prof = NULL;
#ifdef HM355_PROFILE
  HM_CHECK(c, hipMalloc((void **)&P.prof, 2 * HM_PROF_N * sizeof(unsigned long long))); HM_CHECK(c, hipMemset(P.prof, 0, 2 * HM_PROF_N * sizeof(unsigned long long)));
#elif defined(HM355_TRACE)
  HM_CHECK(c, hipMalloc((void **)&P.prof, (1 + 3 * (size_t)HM_TRACE_CAP) * sizeof(unsigned long long))); HM_CHECK(c, hipMemset(P.prof, 0, sizeof(unsigned long long)));
#endif
  HM_CHECK(c, hipMemcpy(c->dP, &P, sizeof(Params), hipMemcpyHostToDevice));
  return HM355_OK;
}

extern "C" void hm355_destroy(hm355_ctx *c)
{
  if (!c) return;
  for (size_t s = 0; s < c->slots.size(); s++) {
    if (c->slots[s].imeta) hipFree(c->slots[s].imeta);
    for (int k = 0; k < 3; k++) if (c->slots[s].saoSrc[k]) hipFree(c->slots[s].saoSrc[k]);
    { Slot &sl = c->slots[s]; if (sl.rawIn) hipFree(sl.rawIn); if (sl.rawOut) hipFree(sl.rawOut); if (sl.bitsRaw) hipFree(sl.bitsRaw); if (sl.bitsPacked) hipFree(sl.bitsPacked); if (sl.bitsSizes) hipFree(sl.bitsSizes);
      if (sl.bitsSync) hipFree(sl.bitsSync); if (sl.bitsFlag) hipFree(sl.bitsFlag); if (sl.bitsIp) hipFree(sl.bitsIp); }
    { Slot &sl = c->slots[s]; if (sl.dDqp) hipFree(sl.dDqp); if (sl.dCtuQp) hipFree(sl.dCtuQp); if (sl.dDqpOut) hipFree(sl.dDqpOut); if (sl.dRowFlag) hipFree(sl.dRowFlag); }
    if (c->slots[s].saoStat) hipFree(c->slots[s].saoStat); if (c->slots[s].saoCand) hipFree(c->slots[s].saoCand); if (c->slots[s].saoCoded) hipFree(c->slots[s].saoCoded); if (c->slots[s].saoRecon) hipFree(c->slots[s].saoRecon);
  }
  for (int l = 0; l < HM_MAX_LANES; l++) {
    Lane &L = c->lane[l];
    if (L.busy && L.stream) hipStreamSynchronize(L.stream);
    if (L.dItems) hipFree(L.dItems);
    if (L.dTeamWin) hipFree(L.dTeamWin);
    if (l == 0) continue;           // lane 0 wraps the context's own objects, released below
    if (L.dWs) hipFree(L.dWs); if (L.dSched) hipFree(L.dSched); if (L.dP) hipFree(L.dP);
    if (L.ev0) hipEventDestroy(L.ev0); if (L.ev1) hipEventDestroy(L.ev1); if (L.stream) hipStreamDestroy(L.stream);
  }
  if (c->arena) hipFree(c->arena);
  if (c->hStage) (void)hipHostFree(c->hStage);
  if (c->dTab) hipFree(c->dTab); if (c->dWs) hipFree(c->dWs); if (c->dFrames) hipFree(c->dFrames); if (c->dP) hipFree(c->dP); if (c->dSched) hipFree(c->dSched);
  if (c->staging) hipHostFree(c->staging);
  if (c->dDbk) hipFree(c->dDbk);
  if (c->dSao) hipFree(c->dSao);
  if (c->dBits) hipFree(c->dBits);
  if (c->dIngest) hipFree(c->dIngest);
  if (c->ev0) hipEventDestroy(c->ev0); if (c->ev1) hipEventDestroy(c->ev1); if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int hm355_upload(hm355_ctx *c, int slot, const hm355_planes *org)
{
  if (!c || !org || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp; FrameBuf &fb = c->slots[slot].fb;
  for (int k = 0; k < 3; k++) {
    if (!org->plane[k]) return fail(c, HM355_ERR_ARG, "null plane");
    const int w = P.width >> (k ? 1 : 0), h = P.height >> (k ? 1 : 0);
    HM_CHECK(c, hipMemcpy2DAsync(fb.org[k], (size_t)P.stride[k] * sizeof(Pel), org->plane[k], (size_t)w * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, c->stream));
  }
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  return HM355_OK;
}

// cu_qp_delta state of a slot for the search about to be launched (hm355_set_dqp armed it): the quantiser parameters of every QP the CTUs can
// take (they depend on the slice's lambda), the CTUs' QPs, m_bEncodeDQP on entry, the per-row assumptions under WaveFrontSynchro
static int dqp_prepare(hm355_ctx *c, int slot, const hm355_slice_desc *sd, hipStream_t stream)
{
  Slot &sl = c->slots[slot]; const Params &P = c->hp;
  sl.lastSlice = *sd;
  if (!sl.dqpOn) { sl.fb.dqp = NULL; return HM355_OK; }
  if (!sl.dDqp) {
    HM_CHECK(c, hipMalloc((void **)&sl.dDqp, sizeof(DqpPic))); HM_CHECK(c, hipMalloc((void **)&sl.dCtuQp, c->numCtus));
    HM_CHECK(c, hipMalloc((void **)&sl.dDqpOut, sizeof(CtuDqp) * c->numCtus)); HM_CHECK(c, hipMalloc((void **)&sl.dRowFlag, P.hCtu));
  }
  DqpPic *hp = new DqpPic; memset(hp, 0, sizeof(*hp));
  hp->flagIn = sl.dqpFlagIn; hp->sliceQp = sd->qp; hp->ctuQp = sl.dCtuQp; hp->out = sl.dDqpOut; hp->rowFlag = sl.dRowFlag;
  for (int q = -12; q <= 51; q++) {
    FrameBuf t; memset(&t, 0, sizeof(t));
    hm355_fill_slice_params(&t, P.bitDepth, q, sd->lambda, sd->chroma_weight);
    QpTab &e = hp->tab[q + 12];
    for (int k = 0; k < 2; k++) { e.qpPer[k] = t.qpPer[k]; e.qpRem[k] = t.qpRem[k]; e.rdFactor[k] = t.rdFactor[k]; for (int l = 0; l < 4; l++) e.errScale[k][l] = t.errScale[k][l]; }
  }
  std::vector<int8_t> q(c->numCtus, (int8_t)sd->qp);
  if (!sl.ctuQp.empty()) q = sl.ctuQp;
  hipError_t e1 = hipMemcpyAsync(sl.dDqp, hp, sizeof(DqpPic), hipMemcpyHostToDevice, stream);
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(sl.dCtuQp, q.data(), c->numCtus, hipMemcpyHostToDevice, stream);
  if (e1 == hipSuccess) e1 = hipMemcpyAsync(sl.dRowFlag, sl.rowFlag.data(), P.hCtu, hipMemcpyHostToDevice, stream);
  if (e1 == hipSuccess) e1 = hipStreamSynchronize(stream);
  delete hp;
  HM_CHECK(c, e1);
  sl.fb.dqp = sl.dDqp;
  return HM355_OK;
}

// Lane l of the context: lane 0 wraps the context's own stream, scratch areas and scheduler words; further lanes get theirs on first use.
static int lane_prepare(hm355_ctx *c, int l)
{
  Lane &L = c->lane[l];
  if (L.stream) return HM355_OK;
  if (l == 0) { L.stream = c->stream; L.ev0 = c->ev0; L.ev1 = c->ev1; L.dP = c->dP; L.dWs = c->dWs; L.wsCount = c->wsCount; L.dSched = c->dSched; L.fewWaves = c->hp.fewWaves; return HM355_OK; }
  HM_CHECK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
  HM_CHECK(c, hipEventCreate(&L.ev0)); HM_CHECK(c, hipEventCreate(&L.ev1));
  L.wsCount = c->wsCount;
  HM_CHECK(c, hipMalloc((void **)&L.dWs, L.wsCount * sizeof(WorkSpace)));
  HM_CHECK(c, hipMalloc((void **)&L.dSched, 64)); HM_CHECK(c, hipMemset(L.dSched, 0, 64));
  HM_CHECK(c, hipMalloc((void **)&L.dP, sizeof(Params)));
  L.fewWaves = -1;
  return HM355_OK;
}

// Enqueues the search over CTU rows [row0, row1] of the pictures in slots [slot0, slot0 + n) on lane l and returns; rows above row0 hold
// finished (or imported) CTUs.  Launches of different lanes run concurrently (each on its own stream with its own scratch areas), so the
// drain of one step overlaps the fill of the next; their slot ranges must not overlap.
static int run_begin(hm355_ctx *c, int l, int slot0, int n, const hm355_slice_desc *slices, int row0, int row1)
{
  const Params &P = c->hp;
  int rc = lane_prepare(c, l);
  if (rc != HM355_OK) return rc;
  Lane &L = c->lane[l];
  if (L.busy) return fail(c, HM355_ERR_ARG, "hm355_run_begin: the lane still has a launch in flight (hm355_run_wait first)");
  L.fbs.resize(n);
  for (int f = 0; f < n; f++) {
    if (slices[f].slice_type != 2) return fail(c, HM355_ERR_ARG, "only I slices are supported");
    if (slices[f].qp < 0 || slices[f].qp > 51 || !(slices[f].lambda > 0) || !(slices[f].chroma_weight > 0)) return fail(c, HM355_ERR_ARG, "bad slice parameters");
    hm355_fill_slice_params(&c->slots[slot0 + f].fb, P.bitDepth, slices[f].qp, slices[f].lambda, slices[f].chroma_weight);
    { const int rc2 = dqp_prepare(c, slot0 + f, slices + f, L.stream); if (rc2 != HM355_OK) return rc2; }
    L.fbs[f] = c->slots[slot0 + f].fb;
  }
  HM_CHECK(c, hipMemcpyAsync(c->dFrames + slot0, L.fbs.data(), sizeof(FrameBuf) * n, hipMemcpyHostToDevice, L.stream));
  int carry = 0;
  for (int f = 0; f < n; f++) carry |= c->slots[slot0 + f].fb.imeta != NULL && (P.height & 63) != 0;
  const long long key[5] = {slot0, n, carry, row0, row1};      // the cached work list is reused only for the very same launch shape
  if (!L.keyValid || memcmp(key, L.key, sizeof(key)) != 0) {
    L.keyValid = 0;
    hm355_build_schedule(P.wCtu, P.hCtu, P.wpp, n, L.items, L.stepStart, carry, slot0, row0, row1);
    if (L.items.size() > L.itemsCap) {
      if (L.dItems) hipFree(L.dItems);
      L.dItems = NULL; L.itemsCap = 0;
      HM_CHECK(c, hipMalloc((void **)&L.dItems, sizeof(WorkItem) * L.items.size()));
      L.itemsCap = L.items.size();
    }
    HM_CHECK(c, hipMemcpyAsync(L.dItems, L.items.data(), sizeof(WorkItem) * L.items.size(), hipMemcpyHostToDevice, L.stream));
    memcpy(L.key, key, sizeof(key)); L.keyValid = 1;
  }
  // A WPP picture offers about 16 CTUs at a time (one when the CABAC state chains through all of them).  A launch that cannot keep ~5
  // one-wavefront searches per CU busy prefers the shortest dependency chain over the fewest instructions (fewWaves); one that cannot
  // even give every CU two searches runs as teams of HM_TEAM wavefronts per CTU (HM355_TEAM=0 / 1 overrides for A/B runs).
  // P / B slices: a team (nine wavefronts, one team per CU) takes a one-stream picture through 3x faster than one wavefront per CTU, and 256 teams together do about half of
  // what 2,816 one-wavefront searches do: measured on 1080p low-delay P streams with WaveFrontSynchro (CTU/s, one wavefront / teams): 32 streams
  // 798 / 1,556, 64: 1,538 / 2,356, 128: 2,862 / 2,944 -- teams up to 96 streams (1,024 when every stream is one serial chain of CTUs).
  const long long parallel = (long long)n * (P.wpp ? 16 : 1);
  const int fewWaves = parallel < 1280 ? 1 : 0;
  int anyInter = 0;
  for (int f = 0; f < n; f++) if (c->slots[slot0 + f].fb.imeta) anyInter = 1;
  int useTeam = (anyInter ? (P.wpp ? n <= 96 : n <= 1024) : parallel <= 512) && L.wsCount >= HM_TEAM;
  { const char *ev = getenv("HM355_TEAM"); if (ev && ev[0] == '0') useTeam = 0; if (ev && ev[0] == '1' && L.wsCount >= HM_TEAM) useTeam = 1; }
  const size_t winSamples = (size_t)65 * P.stride[0] + (size_t)33 * (P.stride[1] + P.stride[2]);
  int teams = 0;
  int waves = anyInter ? HM_TEAM : HM_TEAM_I;   // P / B slices: every chain of candidates on two or three wavefronts (hm355_team.h)
  { const char *ev = getenv("HM355_TEAM_WAVES"); if (ev && atoi(ev) == HM_TEAM_I) waves = HM_TEAM_I; }   // A/B runs (five-wavefront teams on P streams: 1,021 / 1,529 / 1,871 CTU/s in the table above)
  if (useTeam) {
    const size_t total = L.items.size();
    // as many teams as CTUs can ever be ready at once (the wavefront's widest step), a few more so that a finished team finds the next ticket taken
    size_t want = (size_t)maxItemsPerStep(P.wCtu, row1 - row0 + 1 < P.hCtu ? row1 - row0 + 1 : P.hCtu, P.wpp, n) + 2;
    if (want > total) want = total; if (want > 512) want = 512; if (want > L.wsCount / waves) want = L.wsCount / waves;
    if (want > L.teamCap) {
      HM_CHECK(c, hipStreamSynchronize(L.stream));
      if (L.dTeamWin) hipFree(L.dTeamWin);
      L.dTeamWin = NULL; L.teamCap = 0; L.fewWaves = -1;
      if (hipMalloc((void **)&L.dTeamWin, want * HM_TEAM_HELPERS * winSamples * sizeof(Pel)) != hipSuccess) { (void)hipGetLastError(); useTeam = 0; }
      else L.teamCap = want;
    }
    teams = (int)(want < L.teamCap ? want : L.teamCap);
  }
  if (fewWaves != L.fewWaves) {
    Params lp = c->hp; lp.ws = L.dWs; lp.fewWaves = fewWaves; lp.teamWin = L.dTeamWin; lp.teamWinStride = winSamples;
    if (l == 0) { c->hp.fewWaves = fewWaves; c->hp.teamWin = L.dTeamWin; c->hp.teamWinStride = winSamples; }
    HM_CHECK(c, hipMemcpyAsync(L.dP, &lp, sizeof(Params), hipMemcpyHostToDevice, L.stream));
    HM_CHECK(c, hipStreamSynchronize(L.stream));      // lp is a local
    L.fewWaves = fewWaves;
  }
  c->epoch++; if (c->epoch == 0) c->epoch = 1;
  HM_CHECK(c, hipMemsetAsync(L.dSched, 0, 32, L.stream));       // ticket = 0, abort = 0, published CTUs = 0 (+ the counters of diagnostic builds)
  if (row0 > 0)    // the row above the band is complete: its CTUs count as published in this run
    for (int f = 0; f < n; f++) HM_CHECK(c, hipMemsetD32Async((hipDeviceptr_t)(c->slots[slot0 + f].fb.done + (size_t)(row0 - 1) * P.wCtu), (int)c->epoch, P.wCtu, L.stream));
  HM_CHECK(c, hipEventRecord(L.ev0, L.stream));
  const int total = (int)L.items.size();
  if (useTeam && teams > 0) {
    L.grid = teams;
    const size_t lds = HM_TEAM_LDS_BYTES(waves);
    if (!c->teamLdsSet) { HM_CHECK(c, hipFuncSetAttribute((const void *)hm355_ctu_team_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HM_TEAM_LDS_BYTES(HM_TEAM))); c->teamLdsSet = 1; }
    hipLaunchKernelGGL(hm355_ctu_team_kernel, dim3(L.grid), dim3(64 * waves), lds, L.stream, (const Params *)L.dP, (const WorkItem *)L.dItems, total, L.dSched, c->epoch);
  } else {
    L.grid = total < (int)L.wsCount ? total : (int)L.wsCount;
    // A caller that keeps `share` launches in flight (hm355_set_lane_share): each launch only takes its share of the searches the device can hold --
    // a persistent workgroup that waits for a neighbouring CTU keeps its place on the CU, so a launch sized for the whole device would lock the
    // others out until its tickets run out, and the launches would run one after the other
    if (c->laneShare > 1) { const int cap = (int)(3072 * 5 / (4 * c->laneShare)); if (L.grid > cap) L.grid = cap; }
    // workgroups of HM_CTU_WAVES independent searches (wavefronts); a search's workspace is blockIdx * HM_CTU_WAVES + wave < wsCount
    int groups = (L.grid + HM_CTU_WAVES - 1) / HM_CTU_WAVES;
    if (groups > (int)(L.wsCount / HM_CTU_WAVES)) groups = (int)(L.wsCount / HM_CTU_WAVES);
    L.grid = groups * HM_CTU_WAVES;
    hipLaunchKernelGGL(hm355_ctu_kernel, dim3(groups), dim3(64 * HM_CTU_WAVES), 0, L.stream, (const Params *)L.dP, (const WorkItem *)L.dItems, total, L.dSched, c->epoch);
  }
  HM_CHECK(c, hipGetLastError());
  HM_CHECK(c, hipEventRecord(L.ev1, L.stream));
  L.busy = 1;
  return HM355_OK;
}
static int run_begin(hm355_ctx *c, int l, int slot0, int n, const hm355_slice_desc *slices, int row0, int row1);
static int run_wait(hm355_ctx *c, int l, double *kernelMs);
// WaveFrontSynchro with cu_qp_delta: TEncCu::m_bEncodeDQP reaches the first CTU of a row from the LAST CTU of the row above (coding order), which the
// wavefront has not searched yet when that row starts.  The launch ran on an assumption per row (Slot::rowFlag, "clear" to begin with: a CTU with
// any coded block leaves it clear); here the assumptions are checked against what the rows above actually left, and from the first row that was
// started on a wrong one the picture is searched again with the corrected value -- rare, and then repeated for the rows below.
static int dqp_verify_rows(hm355_ctx *c, int l, int slot0, int n)
{
  const Params &P = c->hp;
  if (!P.wpp || P.wCtu < 1) return HM355_OK;
  for (int f = 0; f < n; f++) {
    Slot &sl = c->slots[slot0 + f];
    if (!sl.dqpOn || !sl.fb.dqp) continue;
    std::vector<CtuDqp> out(c->numCtus);
    for (int y = 1; y < P.hCtu; y++) {
      HM_CHECK(c, hipMemcpy(out.data(), sl.dDqpOut, sizeof(CtuDqp) * c->numCtus, hipMemcpyDeviceToHost));
      int bad = -1;
      for (int yy = y; yy < P.hCtu; yy++) if (sl.rowFlag[yy] != out[(size_t)yy * P.wCtu - 1].flagOut) { bad = yy; break; }
      if (bad < 0) break;
      sl.rowFlag[bad] = out[(size_t)bad * P.wCtu - 1].flagOut;
      const hm355_slice_desc sd = sl.lastSlice;
      int rc = run_begin(c, l, slot0 + f, 1, &sd, bad, P.hCtu - 1);
      if (rc == HM355_OK) rc = run_wait(c, l, NULL);
      if (rc != HM355_OK) return rc;
      y = bad;
    }
  }
  return HM355_OK;
}
static int run_wait(hm355_ctx *c, int l, double *kernelMs)
{
  Lane &L = c->lane[l];
  if (!L.busy) return fail(c, HM355_ERR_ARG, "hm355_run_wait: no launch in flight on this lane");
  L.busy = 0;
  HM_CHECK(c, hipStreamSynchronize(L.stream));
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, L.ev0, L.ev1));
  unsigned int sched[2] = {0, 0};
  // on the lane's own stream: a copy on the null stream would wait for the launches of every other lane as well (legacy stream semantics)
  HM_CHECK(c, hipMemcpyAsync(sched, L.dSched, sizeof(sched), hipMemcpyDeviceToHost, L.stream));
  HM_CHECK(c, hipStreamSynchronize(L.stream));
#if defined(HM355_TEAMSTAT)
  { unsigned int ts[8] = {0}; (void)hipMemcpy(ts, L.dSched, sizeof(ts), hipMemcpyDeviceToHost);
    if (ts[3]) fprintf(stderr, "[team] CTUs %u: main wavefront %.1f ms/CTU = 8x8 CUs %.1f + waiting for helpers %.1f + rest; re-searched CUs %u\n",
                       ts[3], ts[4] * 1.28e-3 / ts[3], ts[6] * 1.28e-3 / ts[3], ts[5] * 1.28e-3 / ts[3], ts[7]); }
#endif
  if (sched[1] != 0) return fail(c, HM355_ERR_DEVICE, "scheduler aborted: a dependency wait timed out");
  c->lastKernelMs = ms; c->lastLaunches = 1;
  if (kernelMs) *kernelMs = ms;
  if (L.keyValid && !L.inFixup) {
    L.inFixup = 1;
    const int rc = dqp_verify_rows(c, l, (int)L.key[0], (int)L.key[1]);
    L.inFixup = 0; L.keyValid = 0;          // the fix-up launches reused the lane's work list
    c->lastKernelMs = ms;
    if (rc != HM355_OK) return rc;
  }
  return HM355_OK;
}
static int run_rows_impl(hm355_ctx *c, int slot0, int n, const hm355_slice_desc *slices, int row0, int row1)
{
  const int rc = run_begin(c, 0, slot0, n, slices, row0, row1);
  return rc != HM355_OK ? rc : run_wait(c, 0, NULL);
}

// Pipelined steps: up to HM355_MAX_LANES launches in flight, each over its own slots (declared in include/hm355.h)
extern "C" int hm355_run_begin(hm355_ctx *c, int lane, int first_slot, int n, const hm355_slice_desc *slices)
{
  if (!c || !slices || lane < 0 || lane >= HM_MAX_LANES || n < 1 || first_slot < 0 || first_slot + n > (int)c->slots.size()) return HM355_ERR_ARG;
  for (int f = 0; f < n; f++) if (c->slots[first_slot + f].fb.imeta) return fail(c, HM355_ERR_ARG, "hm355_run_begin: I slices only");
  for (int l = 0; l < HM_MAX_LANES; l++)
    if (l != lane && c->lane[l].busy && c->lane[l].keyValid && first_slot < (int)(c->lane[l].key[0] + c->lane[l].key[1]) && (int)c->lane[l].key[0] < first_slot + n)
      return fail(c, HM355_ERR_ARG, "hm355_run_begin: the slots overlap a launch in flight on another lane");
  return run_begin(c, lane, first_slot, n, slices, 0, c->hp.hCtu - 1);
}
extern "C" int hm355_set_lane_share(hm355_ctx *c, int launches_in_flight)
{
  if (!c || launches_in_flight < 1 || launches_in_flight > HM_MAX_LANES) return HM355_ERR_ARG;
  c->laneShare = launches_in_flight;
  return HM355_OK;
}
extern "C" int hm355_run_wait(hm355_ctx *c, int lane, double *kernel_ms)
{
  if (!c || lane < 0 || lane >= HM_MAX_LANES) return HM355_ERR_ARG;
  return run_wait(c, lane, kernel_ms);
}

// ------------------------------------------------------------------------------------------------
// cu_qp_delta: adaptive QP / rate control hooks of compressSlice
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_set_dqp(hm355_ctx *c, int slot, const hm355_dqp_desc *d)
{
  if (!c || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  Slot &sl = c->slots[slot];
  if (!d || !d->use_dqp) { sl.dqpOn = 0; sl.fb.dqp = NULL; sl.ctuQp.clear(); return HM355_OK; }
  const int lo = -6 * (c->hp.bitDepth - 8);
  sl.ctuQp.clear();
  if (d->ctu_qp) {
    for (int a = 0; a < c->numCtus; a++) if (d->ctu_qp[a] < lo || d->ctu_qp[a] > 51) return fail(c, HM355_ERR_ARG, "hm355_set_dqp: CTU QP out of range");
    sl.ctuQp.assign(d->ctu_qp, d->ctu_qp + c->numCtus);
  }
  sl.dqpOn = 1; sl.dqpFlagIn = d->dqp_flag_in != 0;
  sl.rowFlag.assign(c->hp.hCtu, 0);
  return HM355_OK;
}
extern "C" int hm355_get_dqp(hm355_ctx *c, int slot, int8_t *qp_out, int32_t *dqp_flag_out)
{
  if (!c || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  Slot &sl = c->slots[slot];
  if (!sl.dqpOn || !sl.dDqpOut) return fail(c, HM355_ERR_ARG, "hm355_get_dqp: the slot was not searched with cu_qp_delta");
  std::vector<CtuDqp> out(c->numCtus);
  HM_CHECK(c, hipMemcpy(out.data(), sl.dDqpOut, sizeof(CtuDqp) * c->numCtus, hipMemcpyDeviceToHost));
  if (qp_out) for (int a = 0; a < c->numCtus; a++) for (int z = 0; z < 256; z++) qp_out[(size_t)a * 256 + z] = z < out[a].firstZ ? out[a].refQp : out[a].qp;
  if (dqp_flag_out) *dqp_flag_out = out[c->numCtus - 1].flagOut;
  return HM355_OK;
}
extern "C" int hm355_preanalyze(hm355_ctx *c, int slot, uint64_t *sums)
{
  if (!c || !sums || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  unsigned long long *d = NULL;
  HM_CHECK(c, hipMemcpy(c->dFrames + slot, &c->slots[slot].fb, sizeof(FrameBuf), hipMemcpyHostToDevice));
  HM_CHECK(c, hipMalloc((void **)&d, sizeof(unsigned long long) * 8 * c->numCtus));
  HM_CHECK(c, hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL(hm355_preanalyze_kernel, dim3(c->numCtus), dim3(64), 0, c->stream, (const Params *)c->dP, slot, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(sums, d, sizeof(unsigned long long) * 8 * c->numCtus, hipMemcpyDeviceToHost);
  hipFree(d);
  HM_CHECK(c, e);
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->lastKernelMs = ms; c->lastLaunches = 1;
  return HM355_OK;
}

extern "C" int hm355_run(hm355_ctx *c, int n, const hm355_slice_desc *slices)
{
  if (!c || !slices || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  return run_rows_impl(c, 0, n, slices, 0, c->hp.hCtu - 1);
}

// ------------------------------------------------------------------------------------------------
// CTU-row bands: one device searches rows [first_row, last_row] of its pictures; what the band below needs from the
// band's last row travels through hm355_export_boundary / hm355_import_boundary (host buffers; the transport is the caller's)
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_run_rows(hm355_ctx *c, int first_slot, int n, const hm355_slice_desc *slices, int first_row, int last_row)
{
  if (!c || !slices || n < 1 || first_slot < 0 || first_slot + n > (int)c->slots.size()) return HM355_ERR_ARG;
  if (first_row < 0 || last_row < first_row || last_row >= c->hp.hCtu) return fail(c, HM355_ERR_ARG, "hm355_run_rows: bad row range");
  if (first_row > 0 && !c->hp.wpp) return fail(c, HM355_ERR_ARG, "hm355_run_rows: a band below the first row needs WaveFrontSynchro=1 (without it the CABAC state chains through every CTU)");
  for (int f = 0; f < n; f++) if (c->slots[first_slot + f].fb.imeta) return fail(c, HM355_ERR_ARG, "hm355_run_rows: I slices only");
  return run_rows_impl(c, first_slot, n, slices, first_row, last_row);
}
// bytes of one picture's boundary row: the bottom sample line of the three planes, the decision arrays and the CABAC state after each CTU of the row
extern "C" size_t hm355_boundary_bytes(const hm355_ctx *c)
{
  if (!c) return 0;
  const Params &P = c->hp;
  return (size_t)(P.stride[0] + P.stride[1] + P.stride[2]) * sizeof(Pel) + (size_t)P.wCtu * (sizeof(CtuMeta) + sizeof(Cabac));
}
static int boundary_copy(hm355_ctx *c, int slot, int row, void *buf, int toHost)
{
  if (!c || !buf || slot < 0 || slot >= (int)c->slots.size() || row < 0 || row >= c->hp.hCtu) return HM355_ERR_ARG;
  const Params &P = c->hp; FrameBuf &fb = c->slots[slot].fb;
  uint8_t *p = (uint8_t *)buf;
  const hipMemcpyKind kind = hipMemcpyDefault;       // buf may be host or device memory (a device buffer goes straight to RCCL)
  for (int k = 0; k < 3; k++) {         // the line right above the next CTU row: all that intra prediction reads across the boundary (TComPattern.cpp:107-165)
    Pel *line = fb.rec[k] + ((size_t)(row + 1) * (k ? 32 : 64) - 1) * P.stride[k];
    const size_t bytes = (size_t)P.stride[k] * sizeof(Pel);
    HM_CHECK(c, toHost ? hipMemcpy(p, line, bytes, kind) : hipMemcpy(line, p, bytes, kind));
    p += bytes;
  }
  CtuMeta *meta = fb.meta + (size_t)row * P.wCtu; Cabac *es = fb.endState + (size_t)row * P.wCtu;
  HM_CHECK(c, toHost ? hipMemcpy(p, meta, sizeof(CtuMeta) * P.wCtu, kind) : hipMemcpy(meta, p, sizeof(CtuMeta) * P.wCtu, kind));
  p += sizeof(CtuMeta) * P.wCtu;
  HM_CHECK(c, toHost ? hipMemcpy(p, es, sizeof(Cabac) * P.wCtu, kind) : hipMemcpy(es, p, sizeof(Cabac) * P.wCtu, kind));
  return HM355_OK;
}
extern "C" int hm355_export_boundary(hm355_ctx *c, int slot, int row, void *buf) { return boundary_copy(c, slot, row, buf, 1); }
extern "C" int hm355_import_boundary(hm355_ctx *c, int slot, int row, const void *buf) { return boundary_copy(c, slot, row, (void *)buf, 0); }

#if defined(HM355_TRACE)
// diagnostic build only: copies and resets the RD-evaluation trace (3 words per record), returns the record count
extern "C" long long hm355_read_trace(hm355_ctx *c, unsigned long long *out, long long cap)
{
  unsigned long long n = 0;
  if (hipMemcpy(&n, c->hp.prof, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (n > HM_TRACE_CAP) n = HM_TRACE_CAP;
  if ((long long)n > cap) n = (unsigned long long)cap;
  if (n && hipMemcpy(out, c->hp.prof + 1, n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  hipMemset(c->hp.prof, 0, sizeof(unsigned long long));
  return (long long)n;
}
#endif
#ifdef HM355_PROFILE
extern "C" int hm355_reset_profile(hm355_ctx *c) { return hipMemset(c->hp.prof, 0, 2 * HM_PROF_N * sizeof(unsigned long long)) == hipSuccess ? 0 : HM355_ERR_DEVICE; }
extern "C" int hm355_read_profile(hm355_ctx *c, unsigned long long *out32)
{ return hipMemcpy(out32, c->hp.prof, 2 * HM_PROF_N * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : HM355_ERR_DEVICE; }
#endif

extern "C" int hm355_last_run_info(const hm355_ctx *c, double *kernel_ms, int *launches)
{
  if (!c) return HM355_ERR_ARG;
  if (kernel_ms) *kernel_ms = c->lastKernelMs;
  if (launches) *launches = c->lastLaunches;
  return HM355_OK;
}

extern "C" int hm355_download(hm355_ctx *c, int slot, hm355_planes *rec, hm355_ctu_out *ctus, hm355_slice_stats *stats)
{
  if (!c || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp; FrameBuf &fb = c->slots[slot].fb;
  if (rec) for (int k = 0; k < 3; k++) if (!rec->plane[k]) return fail(c, HM355_ERR_ARG, "null plane");
  // one picture's results cross PCIe into a pinned staging buffer (asynchronous copies on the context's stream, one wait), then go to the caller's
  // pageable buffers with plain memcpy: 81 MB per 4K picture
  const size_t nC = (size_t)c->numCtus, bStat = sizeof(CtuStat) * nC, bMeta = sizeof(CtuMeta) * nC, bCoef = sizeof(TCoeff) * nC * HM_COEF_CTU;
  size_t bPlane[3], oPlane[3], off = (bStat + bMeta + bCoef + 255) & ~(size_t)255;
  for (int k = 0; k < 3; k++) { bPlane[k] = (size_t)(P.width >> (k ? 1 : 0)) * 2 * (size_t)(P.height >> (k ? 1 : 0)); oPlane[k] = off; off += (bPlane[k] + 255) & ~(size_t)255; }
  if (c->hStageBytes < off) {
    if (c->hStage) (void)hipHostFree(c->hStage);
    c->hStage = NULL; c->hStageBytes = 0;
    HM_CHECK(c, hipHostMalloc((void **)&c->hStage, off, hipHostMallocDefault));
    c->hStageBytes = off;
  }
  CtuStat *st = (CtuStat *)c->hStage; CtuMeta *meta = (CtuMeta *)(c->hStage + bStat); TCoeff *coef = (TCoeff *)(c->hStage + bStat + bMeta);
  if (rec)
    for (int k = 0; k < 3; k++) {
      const int w = P.width >> (k ? 1 : 0), h = P.height >> (k ? 1 : 0);
      HM_CHECK(c, hipMemcpy2DAsync(c->hStage + oPlane[k], (size_t)w * 2, fb.rec[k], (size_t)P.stride[k] * sizeof(Pel), (size_t)w * 2, h, hipMemcpyDeviceToHost, c->stream));
    }
  if (ctus || stats) HM_CHECK(c, hipMemcpyAsync(st, fb.stat, bStat, hipMemcpyDeviceToHost, c->stream));
  if (ctus) {
    HM_CHECK(c, hipMemcpyAsync(meta, fb.meta, bMeta, hipMemcpyDeviceToHost, c->stream));
    HM_CHECK(c, hipMemcpyAsync(coef, fb.coef, bCoef, hipMemcpyDeviceToHost, c->stream));
  }
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  if (rec) for (int k = 0; k < 3; k++) memcpy(rec->plane[k], c->hStage + oPlane[k], bPlane[k]);
  if (stats) {
    stats->pic_total_bits = 0; stats->pic_rd_cost = 0; stats->pic_dist = 0;
    for (int a = 0; a < c->numCtus; a++) { stats->pic_total_bits += st[a].bits; stats->pic_rd_cost += st[a].cost; stats->pic_dist += st[a].dist; }
  }
  if (ctus)
    for (int a = 0; a < c->numCtus; a++) {
      hm355_ctu_out *o = ctus + a; const CtuMeta *m = &meta[a];
      o->total_cost = st[a].cost; o->total_bits = st[a].bits; o->total_dist = st[a].dist;
      memcpy(o->depth, m->depth, 256); memcpy(o->part_size, m->part, 256); memcpy(o->pred_mode, m->pred, 256);
      memcpy(o->intra_dir_luma, m->dirL, 256); memcpy(o->intra_dir_chroma, m->dirC, 256); memcpy(o->tr_idx, m->tr, 256);
      memcpy(o->cbf, m->cbf, 768); memcpy(o->tskip, m->ts, 768);
      const TCoeff *cf = coef + (size_t)a * HM_COEF_CTU;
      memcpy(o->coeff_y, cf, 4096 * 4); memcpy(o->coeff_cb, cf + 4096, 1024 * 4); memcpy(o->coeff_cr, cf + 5120, 1024 * 4);
    }
  return HM355_OK;
}

extern "C" int hm355_compress_slices(hm355_ctx *c, int n, const hm355_slice_desc *slices, const hm355_planes *org,
                                     hm355_planes *rec, hm355_ctu_out *const *ctus, hm355_slice_stats *stats)
{
  if (!c || !slices || !org || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  int rc;
  for (int f = 0; f < n; f++) if ((rc = hm355_upload(c, f, org + f)) != HM355_OK) return rc;
  if ((rc = hm355_run(c, n, slices)) != HM355_OK) return rc;
  for (int f = 0; f < n; f++)
    if ((rc = hm355_download(c, f, rec ? rec + f : NULL, ctus ? ctus[f] : NULL, stats ? stats + f : NULL)) != HM355_OK) return rc;
  return HM355_OK;
}

extern "C" int hm355_compress_slice(hm355_ctx *c, const hm355_slice_desc *slice, const hm355_planes *org,
                                    hm355_planes *rec, hm355_ctu_out *ctus, hm355_slice_stats *stats)
{
  hm355_ctu_out *cl[1] = { ctus };
  return hm355_compress_slices(c, 1, slice, org, rec, ctus ? cl : NULL, stats);
}

// ------------------------------------------------------------------------------------------------
// P and B slices: reference pictures are uploaded per call (border-extended like TComPicYuv::extendPicBorder); the n pictures
// of one call are independent of each other (e.g. the current pictures of n streams) and run concurrently
// ------------------------------------------------------------------------------------------------
static_assert(sizeof(hm355_ctu_inter_out) == sizeof(InterMeta), "hm355_ctu_inter_out mirrors InterMeta");
namespace {
struct DevAllocs {                     // device allocations that live for one call
  std::vector<void *> p;
  ~DevAllocs() { for (size_t i = 0; i < p.size(); i++) hipFree(p[i]); }
  template <class T> hipError_t make(T **out, size_t count, const void *src)
  {
    *out = NULL;
    hipError_t e = hipMalloc((void **)out, count * sizeof(T));
    if (e != hipSuccess) return e;
    p.push_back(*out);
    return src ? hipMemcpy(*out, src, count * sizeof(T), hipMemcpyHostToDevice) : hipMemset(*out, 0, count * sizeof(T));
  }
};
}
static hipError_t upload_ref_pic(hm355_ctx *c, const hm355_ref_pic *hp, DevAllocs &da, RefPicDev *out)
{
  const Params &P = c->hp;
  RefPicDev r; memset(&r, 0, sizeof(r));
  hipError_t e = hipSuccess;
  for (int cc = 0; cc < 3 && e == hipSuccess; cc++) {
    const int cw = P.width >> (cc ? 1 : 0), ch = P.height >> (cc ? 1 : 0), mg = HM_REF_MARGIN >> (cc ? 1 : 0), st = cw + 2 * mg;
    std::vector<Pel> buf((size_t)st * (ch + 2 * mg));
    for (int y = -mg; y < ch + mg; y++) {
      const int sy = y < 0 ? 0 : (y >= ch ? ch - 1 : y);
      Pel *d = buf.data() + (size_t)(y + mg) * st + mg;
      const uint16_t *srow = hp->plane[cc] + (size_t)sy * cw;
      for (int x = 0; x < cw; x++) d[x] = (Pel)srow[x];
      for (int x = 1; x <= mg; x++) { d[-x] = (Pel)srow[0]; d[cw - 1 + x] = (Pel)srow[cw - 1]; }
    }
    Pel *dv = NULL; e = da.make(&dv, buf.size(), buf.data());
    r.plane[cc] = dv + (size_t)mg * st + mg; r.stride[cc] = st;
  }
  const size_t np = (size_t)c->numCtus * 256;
  uint8_t *dpm = NULL; if (e == hipSuccess) e = da.make(&dpm, np, hp->pred_mode);
  r.predMode = dpm;
  for (int l = 0; l < 2 && e == hipSuccess; l++) {
    MvD *dm = NULL; int8_t *dr = NULL;
    e = da.make(&dm, np, hp->mv[l]); if (e == hipSuccess) e = da.make(&dr, np, hp->ref_idx[l]);
    r.mv[l] = dm; r.refIdx[l] = dr;
    memcpy(r.refPoc[l], hp->ref_poc[l], sizeof(r.refPoc[l])); memcpy(r.refLT[l], hp->ref_lt[l], sizeof(r.refLT[l]));
  }
  r.poc = hp->poc; r.isLongTerm = hp->long_term;
  *out = r;
  return e;
}
extern "C" int hm355_compress_slices_inter(hm355_ctx *c, int n, const hm355_inter_slice_desc *slices, const hm355_planes *org,
                                           hm355_planes *rec, hm355_ctu_out *const *ctus, hm355_ctu_inter_out *const *ictus, hm355_slice_stats *stats)
{
  if (!c || !slices || !org || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  for (int f = 0; f < n; f++) {
    const hm355_inter_slice_desc *sd = slices + f;
    const int isB = sd->base.slice_type == 0;
    if (sd->base.slice_type != 1 && !isB) return fail(c, HM355_ERR_ARG, "hm355_compress_slices_inter: P (1) or B (0) slices");
    if (sd->num_ref_idx[0] < 1 || sd->num_ref_idx[0] > 16 || (isB ? (sd->num_ref_idx[1] < 1 || sd->num_ref_idx[1] > 16) : sd->num_ref_idx[1] != 0) ||
        sd->max_merge_cand < 1 || sd->max_merge_cand > 5 || (sd->cabac_init_type != 0 && sd->cabac_init_type != 1) ||
        sd->col_ref_idx < 0 || sd->col_ref_idx >= sd->num_ref_idx[(isB && !sd->col_from_l0) ? 1 : 0])
      return fail(c, HM355_ERR_ARG, "bad inter slice parameters");
    for (int l = 0; l < 2; l++) for (int i = 0; i < sd->num_ref_idx[l]; i++) {
      const hm355_ref_pic *hp = sd->ref[l][i];
      if (sd->dev_ref[l][i]) continue;
      if (!hp || !hp->plane[0] || !hp->plane[1] || !hp->plane[2] || !hp->pred_mode || !hp->mv[0] || !hp->mv[1] || !hp->ref_idx[0] || !hp->ref_idx[1])
        return fail(c, HM355_ERR_ARG, "null reference picture data");
    }
  }
  int rc;
  for (int f = 0; f < n; f++) if ((rc = hm355_upload(c, f, org + f)) != HM355_OK) return rc;
  DevAllocs da;
  std::vector<const hm355_ref_pic *> seen; std::vector<RefPicDev> devRefs;     // a picture referenced several times is uploaded once
  std::vector<InterMeta *> dIm(n, (InterMeta *)NULL);
  std::vector<hm355_slice_desc> base(n);
  hipError_t e = hipSuccess;
  for (int f = 0; f < n && e == hipSuccess; f++) {
    const hm355_inter_slice_desc *sd = slices + f;
    InterPic hip; memset(&hip, 0, sizeof(hip));
    hip.sliceType = sd->base.slice_type; hip.poc = sd->poc; hip.numRefIdx[0] = sd->num_ref_idx[0]; hip.numRefIdx[1] = sd->num_ref_idx[1];
    hip.colFromL0 = sd->col_from_l0; hip.colRefIdx = sd->col_ref_idx; hip.tmvp = sd->tmvp; hip.mvdL1Zero = sd->mvd_l1_zero;
    hip.maxMergeCand = sd->max_merge_cand; hip.checkLDC = sd->check_ldc; hip.cabacInitType = sd->cabac_init_type;
    hip.lambdaMotionSAD = sd->lambda_motion_sad; hip.lambdaMotionSSE = sd->lambda_motion_sse;
    for (int l = 0; l < 2; l++) for (int i = 0; i < sd->num_ref_idx[l] && e == hipSuccess; i++) {
      if (sd->dev_ref[l][i]) { hip.ref[l][i] = sd->dev_ref[l][i]->dev; continue; }
      const hm355_ref_pic *hp = sd->ref[l][i];
      size_t k = 0; for (; k < seen.size(); k++) if (seen[k] == hp) break;
      if (k == seen.size()) { RefPicDev r; e = upload_ref_pic(c, hp, da, &r); seen.push_back(hp); devRefs.push_back(r); }
      hip.ref[l][i] = devRefs[k];
    }
    for (int i1 = 0; i1 < sd->num_ref_idx[1]; i1++) {            // TComSlice::setList1IdxToList0Idx
      hip.list1ToList0[i1] = -1;
      for (int i0 = 0; i0 < sd->num_ref_idx[0]; i0++) if (hip.ref[0][i0].poc == hip.ref[1][i1].poc) { hip.list1ToList0[i1] = i0; break; }
    }
    FrameBuf &fb = c->slots[f].fb;
    InterPic *dIp = NULL; MvD *dIntMv = NULL;
    if (e == hipSuccess) e = da.make(&dIp, 1, &hip);
    if (e == hipSuccess && !c->slots[f].imeta) e = hipMalloc((void **)&c->slots[f].imeta, sizeof(InterMeta) * c->numCtus);
    if (e == hipSuccess) { dIm[f] = c->slots[f].imeta; e = hipMemset(dIm[f], 0, sizeof(InterMeta) * c->numCtus); }
    if (e == hipSuccess) e = da.make(&dIntMv, (size_t)c->numCtus * 32, NULL);
    fb.imeta = dIm[f]; fb.ip = dIp; fb.intMv = dIntMv;
    base[f] = sd->base; base[f].slice_type = 2;            // hm355_run validates the common fields
  }
  if (e != hipSuccess) { c->err = std::string("reference picture upload: ") + hipGetErrorString(e); rc = HM355_ERR_DEVICE; }
  else rc = hm355_run(c, n, base.data());
  for (int f = 0; f < n && rc == HM355_OK; f++) {
    rc = hm355_download(c, f, rec ? rec + f : NULL, ctus ? ctus[f] : NULL, stats ? stats + f : NULL);
    if (rc == HM355_OK && ictus && ictus[f] && hipMemcpy(ictus[f], dIm[f], sizeof(InterMeta) * c->numCtus, hipMemcpyDeviceToHost) != hipSuccess) { c->err = "motion download failed"; rc = HM355_ERR_DEVICE; }
  }
  for (int f = 0; f < n; f++) { FrameBuf &fb = c->slots[f].fb; fb.imeta = NULL; fb.ip = NULL; fb.intMv = NULL; }
  return rc;
}
extern "C" int hm355_compress_slice_inter(hm355_ctx *c, const hm355_inter_slice_desc *sd, const hm355_planes *org,
                                          hm355_planes *rec, hm355_ctu_out *ctus, hm355_ctu_inter_out *ictus, hm355_slice_stats *stats)
{
  hm355_ctu_out *cl[1] = { ctus }; hm355_ctu_inter_out *il[1] = { ictus };
  return hm355_compress_slices_inter(c, 1, sd, org, rec, ctus ? cl : NULL, ictus ? il : NULL, stats);
}

// ------------------------------------------------------------------------------------------------
// device-resident reference pictures
// ------------------------------------------------------------------------------------------------
extern "C" void hm355_ref_release(hm355_ctx *c, hm355_ref *r)
{
  (void)c;
  if (!r) return;
  for (size_t i = 0; i < r->owned.size(); i++) hipFree(r->owned[i]);
  delete r;
}
extern "C" int hm355_ref_from_slot(hm355_ctx *c, int slot, int32_t poc, int32_t is_inter, const int32_t num_ref[2], const int32_t ref_poc[2][16],
                                   const int32_t ref_lt[2][16], hm355_ref **out)
{
  if (!c || !out || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  if (is_inter && !c->slots[slot].imeta) return fail(c, HM355_ERR_ARG, "hm355_ref_from_slot: the slot holds no motion data");
  const Params &P = c->hp;
  hm355_ref *r = new hm355_ref(); memset(&r->dev, 0, sizeof(r->dev));
  FrameBuf fbh = c->slots[slot].fb; fbh.imeta = is_inter ? c->slots[slot].imeta : NULL;
  hipError_t e = hipMemcpy(c->dFrames + slot, &fbh, sizeof(FrameBuf), hipMemcpyHostToDevice);
  const size_t np = (size_t)c->numCtus * 256;
  Pel *pl[3] = {NULL, NULL, NULL}; uint8_t *pm = NULL; MvD *mv[2] = {NULL, NULL}; int8_t *ri[2] = {NULL, NULL};
  for (int k = 0; k < 3 && e == hipSuccess; k++) {
    const int cw = P.width >> (k ? 1 : 0), ch = P.height >> (k ? 1 : 0), mg = HM_REF_MARGIN >> (k ? 1 : 0), st = cw + 2 * mg;
    e = hipMalloc((void **)&pl[k], (size_t)st * (ch + 2 * mg) * sizeof(Pel));
    if (e == hipSuccess) { r->owned.push_back(pl[k]); r->dev.plane[k] = pl[k] + (size_t)mg * st + mg; r->dev.stride[k] = st; }
  }
  if (e == hipSuccess) { e = hipMalloc((void **)&pm, np); if (e == hipSuccess) r->owned.push_back(pm); }
  for (int l = 0; l < 2 && e == hipSuccess; l++) {
    e = hipMalloc((void **)&mv[l], np * sizeof(MvD)); if (e == hipSuccess) r->owned.push_back(mv[l]);
    if (e == hipSuccess) { e = hipMalloc((void **)&ri[l], np); if (e == hipSuccess) r->owned.push_back(ri[l]); }
  }
  if (e == hipSuccess) {
    for (int k = 0; k < 3; k++) {
      const int cw = P.width >> (k ? 1 : 0), ch = P.height >> (k ? 1 : 0), mg = HM_REF_MARGIN >> (k ? 1 : 0), st = cw + 2 * mg;
      hipLaunchKernelGGL(hm355_ref_kernel, dim3((st + 63) / 64, ch + 2 * mg, 1), dim3(64, 1, 1), 0, c->stream, c->dP, slot, k, pl[k], pm, mv[0], mv[1], ri[0], ri[1]);
    }
    hipLaunchKernelGGL(hm355_ref_kernel, dim3((unsigned)((np + 63) / 64), 1, 1), dim3(64, 1, 1), 0, c->stream, c->dP, slot, 3, pl[0], pm, mv[0], mv[1], ri[0], ri[1]);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (e != hipSuccess) { c->err = std::string("hm355_ref_from_slot: ") + hipGetErrorString(e); hm355_ref_release(c, r); return HM355_ERR_DEVICE; }
  r->dev.predMode = pm; r->dev.mv[0] = mv[0]; r->dev.mv[1] = mv[1]; r->dev.refIdx[0] = ri[0]; r->dev.refIdx[1] = ri[1];
  r->dev.poc = poc; r->dev.isLongTerm = 0;
  if (ref_poc) memcpy(r->dev.refPoc, ref_poc, sizeof(r->dev.refPoc));
  if (ref_lt) memcpy(r->dev.refLT, ref_lt, sizeof(r->dev.refLT));
  (void)num_ref;
  *out = r;
  return HM355_OK;
}

// A finished reference picture as one blob ("Inter (C5)": pictures of one temporal layer run on different devices, the finished
// pictures are all-gathered): header, the three border-extended planes, the compressed motion field.  buf may be host or device memory,
// so a device buffer goes to RCCL as it is.
struct RefBlobHdr { uint32_t magic, width, height, bitDepth; int32_t stride[3], poc, isLongTerm, refPoc[2][16], refLT[2][16]; double user[4]; };
static size_t ref_plane_bytes(const hm355_ctx *c, int k)
{ const int cw = c->hp.width >> (k ? 1 : 0), ch = c->hp.height >> (k ? 1 : 0), mg = HM_REF_MARGIN >> (k ? 1 : 0); return (size_t)(cw + 2 * mg) * (ch + 2 * mg) * sizeof(Pel); }
extern "C" size_t hm355_ref_bytes(const hm355_ctx *c)
{
  if (!c) return 0;
  const size_t np = (size_t)c->numCtus * 256;
  size_t n = (sizeof(RefBlobHdr) + 255) & ~(size_t)255;
  for (int k = 0; k < 3; k++) n += (ref_plane_bytes(c, k) + 255) & ~(size_t)255;
  return n + ((np + 255) & ~(size_t)255) + 2 * (((np * sizeof(MvD)) + 255) & ~(size_t)255) + 2 * ((np + 255) & ~(size_t)255);
}
static int ref_blob_copy(hm355_ctx *c, hm355_ref *r, uint8_t *p, int toBlob)
{ // the buffers of r in the order hm355_ref_from_slot allocates them: planes Y, Cb, Cr, predMode, mv0, refIdx0, mv1, refIdx1
  const size_t np = (size_t)c->numCtus * 256;
  const size_t sizes[8] = { ref_plane_bytes(c, 0), ref_plane_bytes(c, 1), ref_plane_bytes(c, 2), np, np * sizeof(MvD), np, np * sizeof(MvD), np };
  if (r->owned.size() != 8) return fail(c, HM355_ERR_ARG, "hm355_ref_export / import: not a device-resident reference picture");
  for (int i = 0; i < 8; i++) {
    HM_CHECK(c, toBlob ? hipMemcpy(p, r->owned[i], sizes[i], hipMemcpyDefault) : hipMemcpy(r->owned[i], p, sizes[i], hipMemcpyDefault));
    p += (sizes[i] + 255) & ~(size_t)255;
  }
  return HM355_OK;
}
extern "C" int hm355_ref_export(hm355_ctx *c, const hm355_ref *r, void *buf, const double user[4])
{
  if (!c || !r || !buf) return HM355_ERR_ARG;
  RefBlobHdr h; memset(&h, 0, sizeof(h));
  h.magic = 0x52464d48u; h.width = (uint32_t)c->hp.width; h.height = (uint32_t)c->hp.height; h.bitDepth = (uint32_t)c->hp.bitDepth;
  for (int k = 0; k < 3; k++) h.stride[k] = r->dev.stride[k];
  h.poc = r->dev.poc; h.isLongTerm = r->dev.isLongTerm; memcpy(h.refPoc, r->dev.refPoc, sizeof(h.refPoc)); memcpy(h.refLT, r->dev.refLT, sizeof(h.refLT));
  if (user) memcpy(h.user, user, sizeof(h.user));
  HM_CHECK(c, hipMemcpy(buf, &h, sizeof(h), hipMemcpyDefault));
  return ref_blob_copy(c, (hm355_ref *)r, (uint8_t *)buf + ((sizeof(RefBlobHdr) + 255) & ~(size_t)255), 1);
}
extern "C" int hm355_ref_import(hm355_ctx *c, const void *buf, hm355_ref **out, double user[4])
{
  if (!c || !buf || !out) return HM355_ERR_ARG;
  RefBlobHdr h;
  HM_CHECK(c, hipMemcpy(&h, buf, sizeof(h), hipMemcpyDefault));
  if (h.magic != 0x52464d48u || (int)h.width != c->hp.width || (int)h.height != c->hp.height || (int)h.bitDepth != c->hp.bitDepth)
    return fail(c, HM355_ERR_ARG, "hm355_ref_import: not a reference picture of this sequence");
  const Params &P = c->hp; const size_t np = (size_t)c->numCtus * 256;
  hm355_ref *r = new hm355_ref(); memset(&r->dev, 0, sizeof(r->dev));
  const size_t sizes[8] = { ref_plane_bytes(c, 0), ref_plane_bytes(c, 1), ref_plane_bytes(c, 2), np, np * sizeof(MvD), np, np * sizeof(MvD), np };
  for (int i = 0; i < 8; i++) {
    void *d = NULL;
    if (hipMalloc(&d, sizes[i]) != hipSuccess) { (void)hipGetLastError(); hm355_ref_release(c, r); return fail(c, HM355_ERR_NOMEM, "hm355_ref_import: out of device memory"); }
    r->owned.push_back(d);
  }
  for (int k = 0; k < 3; k++) {
    const int mg = HM_REF_MARGIN >> (k ? 1 : 0), st = (P.width >> (k ? 1 : 0)) + 2 * mg;
    if (h.stride[k] != st) { hm355_ref_release(c, r); return fail(c, HM355_ERR_ARG, "hm355_ref_import: plane layout mismatch"); }
    r->dev.plane[k] = (Pel *)r->owned[k] + (size_t)mg * st + mg; r->dev.stride[k] = st;
  }
  r->dev.predMode = (uint8_t *)r->owned[3]; r->dev.mv[0] = (MvD *)r->owned[4]; r->dev.refIdx[0] = (int8_t *)r->owned[5];
  r->dev.mv[1] = (MvD *)r->owned[6]; r->dev.refIdx[1] = (int8_t *)r->owned[7];
  r->dev.poc = h.poc; r->dev.isLongTerm = h.isLongTerm; memcpy(r->dev.refPoc, h.refPoc, sizeof(h.refPoc)); memcpy(r->dev.refLT, h.refLT, sizeof(h.refLT));
  if (user) memcpy(user, h.user, sizeof(h.user));
  const int rc = ref_blob_copy(c, r, (uint8_t *)buf + ((sizeof(RefBlobHdr) + 255) & ~(size_t)255), 0);
  if (rc != HM355_OK) { hm355_ref_release(c, r); return rc; }
  *out = r;
  return HM355_OK;
}

// ------------------------------------------------------------------------------------------------
// deblocking (TComLoopFilter::loopFilterPic): in place on the reconstruction planes of the slots
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_deblock_run(hm355_ctx *c, int n, const hm355_dbk_desc *descs)
{
  if (!c || !descs || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp;
  std::vector<FrameBuf> fbs(n); std::vector<DbkParams> dps(n);
  for (int f = 0; f < n; f++) {
    if (descs[f].slice_type < 0 || descs[f].slice_type > 2 || descs[f].qp < 0 || descs[f].qp > 51) return fail(c, HM355_ERR_ARG, "bad deblocking parameters");
    if (descs[f].slice_type != 2 && !c->slots[f].imeta) return fail(c, HM355_ERR_ARG, "hm355_deblock_run: the slot holds no motion data (run hm355_compress_slices_inter first)");
    fbs[f] = c->slots[f].fb; fbs[f].imeta = descs[f].slice_type != 2 ? c->slots[f].imeta : NULL;
    dps[f].sliceType = descs[f].slice_type; dps[f].qp = descs[f].qp; memcpy(dps[f].refPoc, descs[f].ref_poc, sizeof(dps[f].refPoc));
  }
  if (!c->dDbk) HM_CHECK(c, hipMalloc((void **)&c->dDbk, sizeof(DbkParams) * c->slots.size()));
  HM_CHECK(c, hipMemcpyAsync(c->dFrames, fbs.data(), sizeof(FrameBuf) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipMemcpyAsync(c->dDbk, dps.data(), sizeof(DbkParams) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  HM_CHECK(c, hipEventRecord(c->ev0, c->stream));
  const int w = P.width, h = P.height;
  const dim3 blk(64, 1, 1);
  // vertical edges (luma, chroma), then horizontal edges: the stream orders the two directions
  hipLaunchKernelGGL(hm355_dbk_kernel, dim3((w / 8 + 63) / 64, h / 4, n), blk, 0, c->stream, c->dP, c->dDbk, 0);
  hipLaunchKernelGGL(hm355_dbk_kernel, dim3((w / 16 + 63) / 64, h / 4, n), blk, 0, c->stream, c->dP, c->dDbk, 1);
  hipLaunchKernelGGL(hm355_dbk_kernel, dim3((w / 4 + 63) / 64, (h / 8 > 1 ? h / 8 : 1), n), blk, 0, c->stream, c->dP, c->dDbk, 2);
  hipLaunchKernelGGL(hm355_dbk_kernel, dim3((w / 4 + 63) / 64, (h / 16 > 1 ? h / 16 : 1), n), blk, 0, c->stream, c->dP, c->dDbk, 3);
  HM_CHECK(c, hipGetLastError());
  HM_CHECK(c, hipEventRecord(c->ev1, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->lastKernelMs = ms; c->lastLaunches = 4;
  return HM355_OK;
}

// ------------------------------------------------------------------------------------------------
// sample adaptive offset (TEncSampleAdaptiveOffset::SAOProcess) on the deblocked pictures of the slots
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_sao_run(hm355_ctx *c, int n, hm355_sao_desc *descs)
{
  if (!c || !descs || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp;
  std::vector<FrameBuf> fbs(n); std::vector<SaoParams> sps(n);
  for (int f = 0; f < n; f++) {
    hm355_sao_desc &d = descs[f];
    if (d.qp < 0 || d.qp > 51 || d.cabac_init_type < 0 || d.cabac_init_type > 2 || d.depth < 0 || d.depth > 7 || !(d.lambda > 0) || !(d.chroma_weight > 0))
      return fail(c, HM355_ERR_ARG, "bad SAO parameters");
    Slot &sl = c->slots[f];
    for (int k = 0; k < 3; k++) if (!sl.saoSrc[k]) HM_CHECK(c, hipMalloc((void **)&sl.saoSrc[k], (size_t)P.stride[k] * P.hCtu * (k ? 32 : 64) * sizeof(Pel)));
    if (!sl.saoStat) HM_CHECK(c, hipMalloc((void **)&sl.saoStat, sizeof(SaoStat) * 3 * c->numCtus));
    if (!sl.saoCand) HM_CHECK(c, hipMalloc((void **)&sl.saoCand, sizeof(SaoCand) * 3 * SAO_NUM_TYPES * c->numCtus));
    if (!sl.saoCoded) HM_CHECK(c, hipMalloc((void **)&sl.saoCoded, sizeof(SaoBlk) * c->numCtus));
    if (!sl.saoRecon) HM_CHECK(c, hipMalloc((void **)&sl.saoRecon, sizeof(SaoBlk) * c->numCtus));
    fbs[f] = sl.fb;
    SaoParams &sp = sps[f]; memset(&sp, 0, sizeof(sp));
    sp.qp = d.qp; sp.cabacInitType = d.cabac_init_type; sp.depth = d.depth;
    sp.lambda[0] = d.lambda; sp.lambda[1] = sp.lambda[2] = d.lambda / d.chroma_weight;
    for (int k = 0; k < 3; k++) { sp.disabledPrev[k] = d.depth > 0 ? d.disabled_rate[k][d.depth - 1] : 0.0; sp.src[k] = sl.saoSrc[k]; }
    sp.stat = sl.saoStat; sp.cand = sl.saoCand; sp.coded = sl.saoCoded; sp.recon = sl.saoRecon;
    for (int k = 0; k < 3; k++)
      HM_CHECK(c, hipMemcpyAsync(sl.saoSrc[k], sl.fb.rec[k], (size_t)P.stride[k] * P.hCtu * (k ? 32 : 64) * sizeof(Pel), hipMemcpyDeviceToDevice, c->stream));
  }
  if (!c->dSao) HM_CHECK(c, hipMalloc((void **)&c->dSao, sizeof(SaoParams) * c->slots.size()));
  HM_CHECK(c, hipMemcpyAsync(c->dFrames, fbs.data(), sizeof(FrameBuf) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipMemcpyAsync(c->dSao, sps.data(), sizeof(SaoParams) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  HM_CHECK(c, hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL(hm355_sao_stats_kernel, dim3(c->numCtus, 3, n), dim3(64, 1, 1), 0, c->stream, c->dP, c->dSao);
  hipLaunchKernelGGL(hm355_sao_decide_kernel, dim3(n, 1, 1), dim3(64, 1, 1), 0, c->stream, c->dP, c->dSao);
  hipLaunchKernelGGL(hm355_sao_apply_kernel, dim3((P.width + 63) / 64, P.height, 3 * n), dim3(64, 1, 1), 0, c->stream, c->dP, c->dSao);
  HM_CHECK(c, hipGetLastError());
  HM_CHECK(c, hipEventRecord(c->ev1, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->lastKernelMs = ms; c->lastLaunches = 3;
  HM_CHECK(c, hipMemcpy(sps.data(), c->dSao, sizeof(SaoParams) * n, hipMemcpyDeviceToHost));
  for (int f = 0; f < n; f++) {
    hm355_sao_desc &d = descs[f];
    for (int k = 0; k < 3; k++) { d.enabled[k] = sps[f].enabled[k]; d.disabled_rate[k][d.depth] = (double)sps[f].numOff[k] / (double)c->numCtus; }
    if (d.params) {
      std::vector<SaoBlk> blk(c->numCtus);
      HM_CHECK(c, hipMemcpy(blk.data(), c->slots[f].saoCoded, sizeof(SaoBlk) * c->numCtus, hipMemcpyDeviceToHost));
      static_assert(sizeof(SaoBlk) == 3 * 35 * 4, "SaoBlk is 3 x (mode, type, aux, offset[32])");
      memcpy(d.params, blk.data(), sizeof(SaoBlk) * c->numCtus);
    }
  }
  return HM355_OK;
}

// host buffers in and out: rec is filtered in place with the CU / TU / motion data of the same slice
extern "C" int hm355_deblock(hm355_ctx *c, const hm355_dbk_desc *desc, const hm355_ctu_out *ctus, const hm355_ctu_inter_out *ictus, hm355_planes *rec)
{
  if (!c || !desc || !ctus || !rec) return HM355_ERR_ARG;
  if (desc->slice_type != 2 && !ictus) return fail(c, HM355_ERR_ARG, "hm355_deblock: inter slices need the motion data");
  const Params &P = c->hp; Slot &sl = c->slots[0]; FrameBuf &fb = sl.fb;
  for (int k = 0; k < 3; k++) {
    if (!rec->plane[k]) return fail(c, HM355_ERR_ARG, "null plane");
    const int w = P.width >> (k ? 1 : 0), h = P.height >> (k ? 1 : 0);
    HM_CHECK(c, hipMemcpy2D(fb.rec[k], (size_t)P.stride[k] * sizeof(Pel), rec->plane[k], (size_t)w * 2, (size_t)w * 2, h, hipMemcpyHostToDevice));
  }
  std::vector<CtuMeta> meta(c->numCtus);
  for (int a = 0; a < c->numCtus; a++) {
    const hm355_ctu_out *o = ctus + a; CtuMeta *m = &meta[a];
    memcpy(m->depth, o->depth, 256); memcpy(m->part, o->part_size, 256); memcpy(m->pred, o->pred_mode, 256);
    memcpy(m->dirL, o->intra_dir_luma, 256); memcpy(m->dirC, o->intra_dir_chroma, 256); memcpy(m->tr, o->tr_idx, 256);
    memcpy(m->cbf, o->cbf, 768); memcpy(m->ts, o->tskip, 768);
  }
  HM_CHECK(c, hipMemcpy(fb.meta, meta.data(), sizeof(CtuMeta) * c->numCtus, hipMemcpyHostToDevice));
  if (desc->slice_type != 2) {
    if (!sl.imeta) HM_CHECK(c, hipMalloc((void **)&sl.imeta, sizeof(InterMeta) * c->numCtus));
    HM_CHECK(c, hipMemcpy(sl.imeta, ictus, sizeof(InterMeta) * c->numCtus, hipMemcpyHostToDevice));
  }
  int rc = hm355_deblock_run(c, 1, desc);
  if (rc != HM355_OK) return rc;
  for (int k = 0; k < 3; k++) {
    const int w = P.width >> (k ? 1 : 0), h = P.height >> (k ? 1 : 0);
    HM_CHECK(c, hipMemcpy2D(rec->plane[k], (size_t)w * 2, fb.rec[k], (size_t)P.stride[k] * sizeof(Pel), (size_t)w * 2, h, hipMemcpyDeviceToHost));
  }
  return HM355_OK;
}

// ------------------------------------------------------------------------------------------------
// picture ingest / output (TVideoIOYuv::read / ::write) between file frames and the slots
// ------------------------------------------------------------------------------------------------
static int ingest_launch(hm355_ctx *c, int n, const std::vector<IngestParams> &ips, int output, int gridW, int gridH)
{
  if (!c->dIngest) HM_CHECK(c, hipMalloc((void **)&c->dIngest, sizeof(IngestParams) * c->slots.size()));
  std::vector<FrameBuf> fbs(n); for (int f = 0; f < n; f++) fbs[f] = c->slots[f].fb;
  HM_CHECK(c, hipMemcpyAsync(c->dFrames, fbs.data(), sizeof(FrameBuf) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipMemcpyAsync(c->dIngest, ips.data(), sizeof(IngestParams) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  HM_CHECK(c, hipEventRecord(c->ev0, c->stream));
  const dim3 grid(((unsigned)((gridW + 7) / 8) * (unsigned)gridH + HM_INGEST_BLOCK - 1) / HM_INGEST_BLOCK, 1, 3 * n);   // sized for the luma plane
  if (output) hipLaunchKernelGGL(hm355_output_kernel, grid, dim3(HM_INGEST_BLOCK), 0, c->stream, c->dP, c->dIngest);
  else hipLaunchKernelGGL(hm355_ingest_kernel, grid, dim3(HM_INGEST_BLOCK), 0, c->stream, c->dP, c->dIngest);
  HM_CHECK(c, hipGetLastError());
  HM_CHECK(c, hipEventRecord(c->ev1, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->lastKernelMs = ms; c->lastLaunches = 1;
  return HM355_OK;
}

extern "C" int hm355_upload_file_frames(hm355_ctx *c, int n, const void *const *frames, int file_width, int file_height, int file_bit_depth)
{
  if (!c || !frames || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp;
  if (file_width < 2 || file_height < 2 || (file_width & 1) || (file_height & 1) || file_width > P.width || file_height > P.height ||
      file_bit_depth < 8 || file_bit_depth > 14) return fail(c, HM355_ERR_ARG, "hm355_upload_file_frames: bad file frame geometry / bit depth");
  const size_t maxBytes = (size_t)P.width * P.height * 3, bytes = (size_t)file_width * file_height * 3 / 2 * (file_bit_depth > 8 ? 2 : 1);
  std::vector<IngestParams> ips(n);
  for (int f = 0; f < n; f++) {
    if (!frames[f]) return fail(c, HM355_ERR_ARG, "null frame");
    Slot &sl = c->slots[f];
    if (!sl.rawIn) HM_CHECK(c, hipMalloc((void **)&sl.rawIn, maxBytes));
    HM_CHECK(c, hipMemcpyAsync(sl.rawIn, frames[f], bytes, hipMemcpyHostToDevice, c->stream));
    ips[f].src = sl.rawIn; ips[f].dst = NULL; ips[f].fileW = file_width; ips[f].fileH = file_height; ips[f].fileBitDepth = file_bit_depth; ips[f].fromOrg = 0;
  }
  return ingest_launch(c, n, ips, 0, P.width, P.height);
}

extern "C" int hm355_download_org(hm355_ctx *c, int slot, hm355_planes *org)
{
  if (!c || !org || slot < 0 || slot >= (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp; FrameBuf &fb = c->slots[slot].fb;
  for (int k = 0; k < 3; k++) {
    if (!org->plane[k]) return fail(c, HM355_ERR_ARG, "null plane");
    const int w = P.width >> (k ? 1 : 0), h = P.height >> (k ? 1 : 0);
    HM_CHECK(c, hipMemcpy2D(org->plane[k], (size_t)w * 2, fb.org[k], (size_t)P.stride[k] * sizeof(Pel), (size_t)w * 2, h, hipMemcpyDeviceToHost));
  }
  return HM355_OK;
}

extern "C" int hm355_download_file_frames(hm355_ctx *c, int n, void *const *frames, int file_bit_depth, int conf_right, int conf_bottom, int source)
{
  if (!c || !frames || n < 1 || n > (int)c->slots.size() || source < 0 || source > 1) return HM355_ERR_ARG;
  const Params &P = c->hp;
  if (conf_right < 0 || conf_bottom < 0 || (conf_right & 1) || (conf_bottom & 1) || conf_right >= P.width || conf_bottom >= P.height ||
      file_bit_depth < 8 || file_bit_depth > 14) return fail(c, HM355_ERR_ARG, "hm355_download_file_frames: bad conformance window / bit depth");
  const int fw = P.width - conf_right, fh = P.height - conf_bottom;
  const size_t maxBytes = (size_t)P.width * P.height * 3, bytes = (size_t)fw * fh * 3 / 2 * (file_bit_depth > 8 ? 2 : 1);
  std::vector<IngestParams> ips(n);
  for (int f = 0; f < n; f++) {
    if (!frames[f]) return fail(c, HM355_ERR_ARG, "null frame");
    Slot &sl = c->slots[f];
    if (!sl.rawOut) HM_CHECK(c, hipMalloc((void **)&sl.rawOut, maxBytes));
    ips[f].src = NULL; ips[f].dst = sl.rawOut; ips[f].fileW = fw; ips[f].fileH = fh; ips[f].fileBitDepth = file_bit_depth; ips[f].fromOrg = source;
  }
  int rc = ingest_launch(c, n, ips, 1, fw, fh);
  if (rc != HM355_OK) return rc;
  for (int f = 0; f < n; f++) HM_CHECK(c, hipMemcpy(frames[f], c->slots[f].rawOut, bytes, hipMemcpyDeviceToHost));
  return HM355_OK;
}

// ------------------------------------------------------------------------------------------------
// bitstream pass (TEncSlice::encodeSlice) on the pictures of the slots
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_num_substreams(const hm355_ctx *c) { return c ? (c->hp.wpp ? c->hp.hCtu : 1) : HM355_ERR_ARG; }

extern "C" int hm355_encode_slices_run(hm355_ctx *c, int n, hm355_bits_desc *descs)
{
  if (!c || !descs || n < 1 || n > (int)c->slots.size()) return HM355_ERR_ARG;
  const Params &P = c->hp;
  const int numSub = P.wpp ? P.hCtu : 1;
  const size_t rawBytes = (size_t)c->numCtus * HM_BITS_CAP_PER_CTU;
  std::vector<FrameBuf> fbs(n); std::vector<BitsParams> bps(n);
  c->epoch++; if (c->epoch == 0) c->epoch = 1;          // 0 is what a fresh flag array holds
  for (int f = 0; f < n; f++) {
    hm355_bits_desc &d = descs[f];
    if (d.slice_type < 0 || d.slice_type > 2 || d.qp < 0 || d.qp > 51 || !d.out || !d.sub_sizes) return fail(c, HM355_ERR_ARG, "bad bitstream pass parameters");
    Slot &sl = c->slots[f];
    if (d.slice_type != 2) {
      if (!sl.imeta) return fail(c, HM355_ERR_ARG, "hm355_encode_slices_run: the slot holds no motion data (run hm355_compress_slices_inter first)");
      if (d.cabac_init_type < 0 || d.cabac_init_type > 1 || d.num_ref_idx[0] < 1 || d.num_ref_idx[0] > 16 || d.num_ref_idx[1] < 0 || d.num_ref_idx[1] > 16 ||
          d.max_merge_cand < 1 || d.max_merge_cand > 5) return fail(c, HM355_ERR_ARG, "bad inter slice header values");
    }
    if ((d.sao_enabled[0] || d.sao_enabled[1]) && !sl.saoCoded) return fail(c, HM355_ERR_ARG, "hm355_encode_slices_run: the slot holds no SAO parameters (run hm355_sao_run first)");
    if (!sl.bitsRaw) HM_CHECK(c, hipMalloc((void **)&sl.bitsRaw, rawBytes));
    if (!sl.bitsPacked) HM_CHECK(c, hipMalloc((void **)&sl.bitsPacked, rawBytes));
    if (!sl.bitsSizes) HM_CHECK(c, hipMalloc((void **)&sl.bitsSizes, sizeof(uint32_t) * P.hCtu));
    if (!sl.bitsSync) HM_CHECK(c, hipMalloc((void **)&sl.bitsSync, sizeof(CabacW) * P.hCtu));
    if (!sl.bitsFlag) { HM_CHECK(c, hipMalloc((void **)&sl.bitsFlag, sizeof(uint32_t) * P.hCtu)); HM_CHECK(c, hipMemset(sl.bitsFlag, 0, sizeof(uint32_t) * P.hCtu)); }
    fbs[f] = sl.fb; fbs[f].imeta = NULL; fbs[f].ip = NULL;
    if (d.slice_type != 2) {
      if (!sl.bitsIp) HM_CHECK(c, hipMalloc((void **)&sl.bitsIp, sizeof(InterPic)));
      std::vector<InterPic> ipv(1); InterPic &ip = ipv[0];   // only the slice header values the PU syntax reads
      memset(&ip, 0, sizeof(ip));
      ip.sliceType = d.slice_type; ip.numRefIdx[0] = d.num_ref_idx[0]; ip.numRefIdx[1] = d.slice_type == 0 ? d.num_ref_idx[1] : 0;
      ip.mvdL1Zero = d.mvd_l1_zero; ip.maxMergeCand = d.max_merge_cand; ip.cabacInitType = d.cabac_init_type;
      HM_CHECK(c, hipMemcpy(sl.bitsIp, &ip, sizeof(InterPic), hipMemcpyHostToDevice));
      fbs[f].imeta = sl.imeta; fbs[f].ip = sl.bitsIp;
    }
    BitsParams &bp = bps[f]; memset(&bp, 0, sizeof(bp));
    bp.sliceType = d.slice_type; bp.qp = d.qp; bp.cabacInitType = d.cabac_init_type;
    bp.saoEnabled[0] = d.sao_enabled[0]; bp.saoEnabled[1] = bp.saoEnabled[2] = d.sao_enabled[1];
    bp.sao = (d.sao_enabled[0] || d.sao_enabled[1]) ? (const int32_t *)sl.saoCoded : NULL;
    bp.raw = sl.bitsRaw; bp.capPerCtu = HM_BITS_CAP_PER_CTU; bp.packed = sl.bitsPacked; bp.subSizes = sl.bitsSizes;
    bp.sync = sl.bitsSync; bp.syncFlag = sl.bitsFlag; bp.sched = c->dSched + 8; bp.epoch = c->epoch; bp.nextInitType = d.slice_type;
    HM_CHECK(c, hipMemsetAsync(sl.bitsSizes, 0, sizeof(uint32_t) * P.hCtu, c->stream));     // a substream nobody coded (abandoned launch) has length 0
  }
  if (!c->dBits) HM_CHECK(c, hipMalloc((void **)&c->dBits, sizeof(BitsParams) * c->slots.size()));
  HM_CHECK(c, hipMemcpyAsync(c->dFrames, fbs.data(), sizeof(FrameBuf) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipMemcpyAsync(c->dBits, bps.data(), sizeof(BitsParams) * n, hipMemcpyHostToDevice, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  HM_CHECK(c, hipEventRecord(c->ev0, c->stream));
  // persistent grid, items handed out by ticket in dependency order (hm355_bits_kernel): any grid size drains, so the only bound is the number
  // of per-workgroup scratch areas
  HM_CHECK(c, hipMemsetAsync(c->dSched + 8, 0, 8, c->stream));          // ticket = 0, abort = 0 for this launch
  const int total = numSub * n;
  const int grid = total < (int)c->wsCount ? total : (int)c->wsCount;
  hipLaunchKernelGGL(hm355_bits_kernel, dim3(grid), dim3(64), 0, c->stream, c->dP, c->dBits, n, c->dSched + 8);
  hipLaunchKernelGGL(hm355_bits_pack_kernel, dim3(numSub, n), dim3(64), 0, c->stream, c->dP, c->dBits, (const unsigned int *)(c->dSched + 8));
  HM_CHECK(c, hipGetLastError());
  HM_CHECK(c, hipEventRecord(c->ev1, c->stream));
  HM_CHECK(c, hipStreamSynchronize(c->stream));
  float ms = 0; HM_CHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->lastKernelMs = ms; c->lastLaunches = 2;
  HM_CHECK(c, hipMemcpy(bps.data(), c->dBits, sizeof(BitsParams) * n, hipMemcpyDeviceToHost));
  unsigned int bsched[2] = {0, 0};
  HM_CHECK(c, hipMemcpy(bsched, c->dSched + 8, sizeof(bsched), hipMemcpyDeviceToHost));
  if (bsched[1]) return fail(c, HM355_ERR_DEVICE, "bitstream pass: a WPP hand-off wait was abandoned");
  for (int f = 0; f < n; f++) {
    hm355_bits_desc &d = descs[f]; Slot &sl = c->slots[f];
    if (bps[f].overflow) return fail(c, HM355_ERR_DEVICE, "bitstream pass: a substream outgrew its buffer");
    HM_CHECK(c, hipMemcpy(d.sub_sizes, sl.bitsSizes, sizeof(uint32_t) * numSub, hipMemcpyDeviceToHost));
    size_t tot = 0; for (int k = 0; k < numSub; k++) tot += d.sub_sizes[k];
    if (tot > d.out_cap) return fail(c, HM355_ERR_ARG, "hm355_encode_slices_run: out_cap too small");
    if (tot) HM_CHECK(c, hipMemcpy(d.out, sl.bitsPacked, tot, hipMemcpyDeviceToHost));
    d.next_cabac_init_type = bps[f].nextInitType; d.num_bins = bps[f].bins;
  }
  return HM355_OK;
}

// host buffers in: the CTU data of one slice goes to slot 0 first
extern "C" int hm355_encode_slice(hm355_ctx *c, hm355_bits_desc *desc, const hm355_ctu_out *ctus, const hm355_ctu_inter_out *ictus, const int32_t *sao)
{
  if (!c || !desc || !ctus) return HM355_ERR_ARG;
  if (desc->slice_type != 2 && !ictus) return fail(c, HM355_ERR_ARG, "hm355_encode_slice: inter slices need the motion data");
  if ((desc->sao_enabled[0] || desc->sao_enabled[1]) && !sao) return fail(c, HM355_ERR_ARG, "hm355_encode_slice: SAO enabled without parameters");
  Slot &sl = c->slots[0]; FrameBuf &fb = sl.fb;
  std::vector<CtuMeta> meta(c->numCtus); std::vector<TCoeff> coef((size_t)c->numCtus * HM_COEF_CTU);
  for (int a = 0; a < c->numCtus; a++) {
    const hm355_ctu_out *o = ctus + a; CtuMeta *m = &meta[a];
    memcpy(m->depth, o->depth, 256); memcpy(m->part, o->part_size, 256); memcpy(m->pred, o->pred_mode, 256);
    memcpy(m->dirL, o->intra_dir_luma, 256); memcpy(m->dirC, o->intra_dir_chroma, 256); memcpy(m->tr, o->tr_idx, 256);
    memcpy(m->cbf, o->cbf, 768); memcpy(m->ts, o->tskip, 768);
    memcpy(&coef[(size_t)a * HM_COEF_CTU], o->coeff_y, 4096 * 4); memcpy(&coef[(size_t)a * HM_COEF_CTU + 4096], o->coeff_cb, 1024 * 4);
    memcpy(&coef[(size_t)a * HM_COEF_CTU + 5120], o->coeff_cr, 1024 * 4);
  }
  HM_CHECK(c, hipMemcpy(fb.meta, meta.data(), sizeof(CtuMeta) * c->numCtus, hipMemcpyHostToDevice));
  HM_CHECK(c, hipMemcpy(fb.coef, coef.data(), sizeof(TCoeff) * coef.size(), hipMemcpyHostToDevice));
  if (desc->slice_type != 2) {
    if (!sl.imeta) HM_CHECK(c, hipMalloc((void **)&sl.imeta, sizeof(InterMeta) * c->numCtus));
    HM_CHECK(c, hipMemcpy(sl.imeta, ictus, sizeof(InterMeta) * c->numCtus, hipMemcpyHostToDevice));
  }
  if (sao) {
    if (!sl.saoCoded) HM_CHECK(c, hipMalloc((void **)&sl.saoCoded, sizeof(SaoBlk) * c->numCtus));
    HM_CHECK(c, hipMemcpy(sl.saoCoded, sao, sizeof(SaoBlk) * c->numCtus, hipMemcpyHostToDevice));
  }
  return hm355_encode_slices_run(c, 1, desc);
}

// ------------------------------------------------------------------------------------------------
// primitive batches
// ------------------------------------------------------------------------------------------------
extern "C" int hm355_dist_batch(hm355_ctx *c, int kind, int n, int bit_depth, int count, const int16_t *org, const int16_t *cur, uint32_t *out)
{
  if (!c || !org || !cur || !out || count < 1 || kind < 0 || kind > 3 || (n != 4 && n != 8 && n != 16 && n != 32 && n != 64) || (bit_depth != 8 && bit_depth != 10)) return HM355_ERR_ARG;
  const size_t bytes = (size_t)count * n * n * sizeof(Pel);
  Pel *dO = NULL, *dC = NULL; uint32_t *dOut = NULL;
  int rc = HM355_OK;
  if (hipMalloc((void **)&dO, bytes) != hipSuccess || hipMalloc((void **)&dC, bytes) != hipSuccess || hipMalloc((void **)&dOut, (size_t)count * 4) != hipSuccess) rc = HM355_ERR_NOMEM;
  if (rc == HM355_OK && (hipMemcpy(dO, org, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dC, cur, bytes, hipMemcpyHostToDevice) != hipSuccess)) rc = HM355_ERR_DEVICE;
  if (rc == HM355_OK) {
    const int grid = count < 65536 ? count : 65536;
    hipLaunchKernelGGL(hm355_dist_kernel, dim3(grid), dim3(64), 0, c->stream, kind, n, bit_depth, count, (const Pel *)dO, (const Pel *)dC, dOut);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, dOut, (size_t)count * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = HM355_ERR_DEVICE; c->err = "dist kernel failed"; }
  }
  if (dO) hipFree(dO); if (dC) hipFree(dC); if (dOut) hipFree(dOut);
  return rc;
}

extern "C" int hm355_transform_batch(hm355_ctx *c, int inverse, int n, int bit_depth, int use_dst, int count, const int32_t *in, int32_t *out)
{
  if (!c || !in || !out || count < 1 || (n != 4 && n != 8 && n != 16 && n != 32) || (bit_depth != 8 && bit_depth != 10) || (use_dst && n != 4)) return HM355_ERR_ARG;
  const size_t bytes = (size_t)count * n * n * 4;
  int32_t *dI = NULL, *dOut = NULL; int rc = HM355_OK;
  if (hipMalloc((void **)&dI, bytes) != hipSuccess || hipMalloc((void **)&dOut, bytes) != hipSuccess) rc = HM355_ERR_NOMEM;
  if (rc == HM355_OK && hipMemcpy(dI, in, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = HM355_ERR_DEVICE;
  if (rc == HM355_OK) {
    const int grid = count < 65536 ? count : 65536;
    hipLaunchKernelGGL(hm355_transform_kernel, dim3(grid), dim3(64), 0, c->stream, inverse, n, bit_depth, use_dst, count, (const int32_t *)dI, dOut);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, dOut, bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = HM355_ERR_DEVICE; c->err = "transform kernel failed"; }
  }
  if (dI) hipFree(dI); if (dOut) hipFree(dOut);
  return rc;
}
