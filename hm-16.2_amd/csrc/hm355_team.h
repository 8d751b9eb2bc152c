// hm355 -- a team of wavefronts searching ONE CTU (latency mode of the CTU search).
//
// TEncCu::xCompressCU evaluates the unsplit CU at depth d and then the four sub-CUs at depth d + 1; both start from the same CABAC
// snapshot ([d][CI_CURR_BEST], TEncCu.cpp:1007-1014 vs :636-798) and read nothing the other one writes: inside the CU each reads only
// what it has reconstructed itself, outside the CU both read the finished neighbours.  The same holds for the two part sizes of the
// smallest CU (2Nx2N and NxN, TEncCu.cpp:706-735).  A workgroup of HM_TEAM wavefronts uses that:
//   * the MAIN wavefront (wave 0) runs the reference's recursion, but hands the unsplit candidate of every depth to a HELPER
//     wavefront (wave 1 + d: the 2Nx2N CU of depth d = 0, 1, 2; wave 4: the 2Nx2N candidate of an 8x8 CU) and goes straight on into
//     the sub-CUs (the NxN candidate at depth 3); it collects the helper's result where the reference compares the two
//     (xCheckBestMode, TEncCu.cpp:1702) -- same operands, same strict "<", same order, so the decision is the reference's;
//   * a helper works on private copies of everything the candidate writes: its own LDS state (Shared), HBM workspace, coefficient
//     area and a window of the reconstruction (the CTU plus the sample row above and the column left of it, same addressing as the
//     picture), filled from the picture when the request arrives.  The main wavefront only writes inside the CU while the helper
//     is busy, the helper only reads outside it before it has written there itself: no hand-shake beyond request / done;
//   * the winner's decision arrays, coefficients, reconstruction and CABAC state are taken from the helper's workspace where the
//     reference copies them from its "best" buffers (TComDataCU::copyToPic, TEncCu::xCopyYuv2Pic, TEncCu.cpp:1087-1110).
// The exact work skipping of the one-wavefront search (hm355_core.h, compress_ctu) needs the unsplit cost before the sub-CUs
// start; here it is applied whenever the helper happens to have answered already -- it never changes a result, so the answer's
// timing does not matter.
// The wavefronts of a team share one CU, so the request / done words live in LDS and workgroup-scope fences order the
// HBM traffic between them.
// P / B slices (compress_ctu_team_inter): helper d evaluates the whole candidate chain of the unsplit CU of depth d = 0, 1, 2
// (compress_cu_inter_modes: merge, 2Nx2N, Nx2N, 2NxN, AMP, intra), the main wavefront the 8x8 CUs.  Two things the reference chains
// through the quadtree cross the wavefronts:
//   * TEncSearch::m_integerMv2Nx2N (TEncSearch.cpp:3880-3888): every 2Nx2N integer search starts from the result of the previous one
//     in evaluation order.  The unsplit CU's own 2Nx2N search depends on nothing else of its candidate chain, so the main wavefront
//     repeats just that search (me_token_prepass) after handing the helper the state before it, and goes on with the state after it;
//   * deriveTestModeAMP (TEncCu.cpp:386-447) gives a sub-CU the part size of its parent's best unsplit mode, which the helper has not
//     found yet when the sub-CUs start.  It only matters when that part size is an AMP one (then merge-only AMP candidates are added
//     where the sub-CU's own best mode does not ask for them): the sub-CUs are searched as if it were not, every candidate chain
//     reports whether an AMP parent would have changed its candidate list (CuFrame::ampSens), and when the parent's answer is AMP and
//     a sub-CU was sensitive, the four sub-CUs are searched again with the part size known -- nothing outside the CU has seen them yet.
#pragma once

#define HM_TEAM 9                      /* wavefronts (and workspaces) of a team in a launch with P / B slices */
#define HM_TEAM_I 5                    /* ... in a launch of I slices only: waves 0..4 */
#define HM_TEAM_HELPERS 4              /* waves 1..4 own a reconstruction window (they evaluate intra candidates) */
#define HM_TEAM_TIMEOUT_TICKS (20ull * 100000000ull)   /* 20 s of the 100 MHz wall clock: a team member that never answers abandons the launch */
// wave 0: main;  1 + d: the unsplit CU of depth d = 0, 1, 2;  4: I slice: the 2Nx2N candidate of an 8x8 CU, P / B slice: the main wavefront's partner
// for the 8x8 CUs;  5 + d: partner of wave 1 + d (P / B slices);  8: second partner of the main wavefront.  A partner takes the Nx2N / 2NxN and the
// vertical AMP candidates of a CU; with two partners the first takes Nx2N, the second 2NxN.
enum { TK_INTRA = 0, TK_CHAIN = 1, TK_PAIR1 = 2, TK_PAIR2 = 3 };

struct TeamBox {                       // mailbox of one helper wavefront (box[w - 1] of wave w)
  uint32_t reqSeq, doneSeq;            // requester: arguments, release, reqSeq + 1;  helper: results, release, doneSeq = reqSeq
  int32_t cuZ, depth, part, kind;      // part: TK_INTRA: the part size to evaluate; TK_CHAIN: the parent's part size (deriveTestModeAMP); TK_PAIR1: bit 0 Nx2N, bit 1 2NxN; TK_PAIR2: bit 0 full, bit 1 merge-only vertical AMP
  int32_t src, sens, owner, improved;  // src: the requesting wave; answers: CuFrame::ampSens (TK_PAIR*: the fractional bits the go-on coder was left with), the wave whose workspace holds the best mode, TK_PAIR*: a candidate came in below the threshold
  uint32_t bits, dist; double cost;    // the candidate as xCheckBestMode sees it (TK_INTRA / TK_CHAIN: split flag of the unsplit CU included)
  double threshold;                    // TK_PAIR2: the best cost so far
};
struct Team {
  TeamBox box[HM_TEAM - 1];
  uint32_t quit, dead;                 // quit: the launch is over; dead: a wait timed out, results are void (the host sees the abort word)
  unsigned int *abortWord;
  WorkItem item;
  int32_t waves, pad_[3];              // wavefronts of this launch's teams: HM_TEAM_I (no partners: a candidate chain stays on one wavefront) or HM_TEAM
  Shared sh[1];                        // [waves of the launch]: the launch sizes the workgroup's LDS for them
};
static_assert(offsetof(Team, sh) % 16 == 0, "Shared holds doubles and 16-byte rows");
#define HM_TEAM_LDS_BYTES(waves) (offsetof(Team, sh) + (size_t)(waves) * sizeof(Shared))

extern __shared__ __align__(16) unsigned char g_team_lds[];   // only the team kernel (hm355_ctu_team_kernel) reaches it
#define HM_TEAM_PTR() ((Team *)g_team_lds)
__device__ __forceinline__ uint32_t team_ld(const uint32_t *p) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void team_st(uint32_t *p, uint32_t v) { if (hm_lane() == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define HM_TEAM_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#define HM_TEAM_ACQUIRE() do { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

#if defined(HM355_TEAMSTAT)            /* diagnostic build (tools/): where the main wavefront's time goes, summed over the CTUs of a launch into the scheduler words */
#define HM_TSTAT_ADD(k, v) do { if (hm_lane() == 0) atomicAdd((unsigned int *)HM_TEAM_PTR()->abortWord + 2 + (k), (unsigned int)(v)); } while (0)
#define HM_TSTAT_T0(t) const unsigned long long t = wall_clock64()
#define HM_TSTAT_T1(k, t) HM_TSTAT_ADD(k, (wall_clock64() - (t)) >> 7)
#else
#define HM_TSTAT_ADD(k, v) ((void)0)
#define HM_TSTAT_T0(t) ((void)0)
#define HM_TSTAT_T1(k, t) ((void)0)
#endif
// ---- main wavefront ----
HM_DEV inline void team_post(Shared *e, int h, int cuZ, int depth, int part, int kind = -1, int src = 0, double threshold = 0.0)
{ // h: box index (wave - 1); kind -1: TK_INTRA on an I slice, TK_CHAIN on a P / B slice
  TeamBox *b = &HM_TEAM_PTR()->box[h];
  if (kind < 0) kind = e->im ? TK_CHAIN : TK_INTRA;
  if (hm_lane() == 0) { b->cuZ = cuZ; b->depth = depth; b->part = part; b->kind = kind; b->src = src; b->threshold = threshold; }
  const uint32_t seq = team_ld(&b->reqSeq) + 1u;
  HM_TEAM_RELEASE();
  team_st(&b->reqSeq, seq);
}
HM_DEV inline int team_ready(Shared *e, int h)
{
  TeamBox *b = &HM_TEAM_PTR()->box[h];
  if (team_ld(&b->doneSeq) != team_ld(&b->reqSeq)) return 0;
  HM_TEAM_ACQUIRE();
  return 1;
}
HM_DEV HM_NOINLINE void team_wait(Shared *e, int h)
{
  HM_ENTRY(e); h = HM_UNI(h);
  Team *T = HM_TEAM_PTR(); TeamBox *b = &T->box[h];
  const uint32_t want = team_ld(&b->reqSeq);
  const unsigned long long t0 = wall_clock64();
  while (team_ld(&b->doneSeq) != want) {
    __builtin_amdgcn_s_sleep(2);
    if (team_ld(&T->dead)) break;
    if (wall_clock64() - t0 > HM_TEAM_TIMEOUT_TICKS) {
      team_st(&T->dead, 1u);
      if (hm_lane() == 0) __hip_atomic_store((__attribute__((address_space(1))) unsigned int *)T->abortWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  HM_TEAM_ACQUIRE();
}
HM_DEV inline WorkSpace *team_helper_ws(Shared *e, int h) { return e->ws + 1 + h; }      // the workspaces of a team lie side by side, the main wavefront's first (called by the main wavefront)
HM_DEV inline WorkSpace *team_wave_ws(int wave) { return HM_TEAM_PTR()->sh[0].ws + wave; }
// the workspace that holds the answer of helper h (box index): its own or its partner's
HM_DEV inline WorkSpace *team_answer_ws(Shared *e, int h) { return team_wave_ws(HM_UNI(HM_TEAM_PTR()->box[h].owner)); }

// what the reference's xCheckBestMode does when the unsplit candidate is the best mode: results of the helper become the CU's
HM_DEV inline void team_take_unsplit(Shared *e, int h, int cuZ, int cuDepth)
{
  WorkSpace *hw = e->im ? team_answer_ws(e, h) : team_helper_ws(e, h);
  restore_best_from(e, &hw->best[cuDepth], cuZ, cuDepth);
  cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &hw->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)]);
}

// TEncCu::compressCtu -> xCompressCU for an I slice, the unsplit candidates evaluated by the helpers (see the head of this file)
HM_DEV HM_NOINLINE void compress_ctu_team(Shared *e)
{
  HM_ENTRY(e);
  CtuMeta *m = (&e->meta);
  CuFrame *fr = e->cuf; int sp = 0;
  fr[0].cuZ = 0; fr[0].phase = 0; fr[0].parentPart = SIZE_NONE;
  double retCost = 0; uint32_t retBits = 0, retDist = 0;
  int pending[4] = {0, 0, 0, 0};       // the unsplit candidate of this depth is with its helper
  while (sp >= 0) {
    CuFrame *f = &fr[sp]; const int cuDepth = sp, cuZ = f->cuZ;
    const int size = 64 >> cuDepth, parts = 256 >> (2 * cuDepth), q = parts >> 2;
    if (f->phase == 0) {
      const int r = hm_z2r(cuZ);
      const int lx = e->ctuX * 64 + (r & 15) * 4, ty = e->ctuY * 64 + (r >> 4) * 4;
      f->boundary = !((lx + size - 1 < e->width) && (ty + size - 1 < e->height));
      f->bestCost = HM_MAX_DOUBLE; f->bestBits = 0; f->bestDist = 0;
      pending[cuDepth] = 0;
      if (!f->boundary) {
        team_post(e, cuDepth, cuZ, cuDepth, SIZE_2Nx2N);
        pending[cuDepth] = 1;
        if (cuDepth == 3) {
          // the helper has the 2Nx2N candidate; the NxN one runs here.  xCheckBestMode sees 2Nx2N first, NxN has to be strictly cheaper.
          check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN);
          const double cN = e->outCost; const uint32_t bN = e->outBits, dN = e->outDist;
          team_wait(e, 3); pending[3] = 0;
          const TeamBox *b = &HM_TEAM_PTR()->box[3];
          f->bestCost = b->cost; f->bestBits = b->bits; f->bestDist = b->dist;
          if (cN < f->bestCost) {
            f->bestCost = cN; f->bestBits = bN; f->bestDist = dN;
            save_best(e, cuZ, cuDepth);
            cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
            restore_best(e, cuZ, cuDepth);
          } else team_take_unsplit(e, 3, cuZ, cuDepth);
          reset_bits(&e->cur);           // TEncCu.cpp:859-863 at the smallest CU size: no split flag, the bit counter is reset all the same
          f->bestCost = calc_rd_cost(e, f->bestBits, f->bestDist);
          retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue;
        }
      }
      init_est_data(e, cuZ, cuDepth);
      f->splitBits = 0; f->splitDist = 0; f->sub = 0; f->phase = 1;
    }
    if (f->phase == 1) {
      if (f->sub < 4) {
        const int s = f->sub++;
        const int subZ = cuZ + s * q, r = hm_z2r(subZ);
        const int sx = e->ctuX * 64 + (r & 15) * 4, sy = e->ctuY * 64 + (r >> 4) * 4;
        HM_PAR_FOR(i, q) { m->depth[subZ + i] = (uint8_t)(cuDepth + 1); m->part[subZ + i] = SIZE_NONE; m->pred[subZ + i] = MODE_NONE; }
        HM_SYNC();
        if (sx < e->width && sy < e->height) {
          if (s == 0) cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
          else cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth + 1, CI_NEXT_BEST)]);
          fr[sp + 1].cuZ = (int16_t)subZ; fr[sp + 1].phase = 0; fr[sp + 1].parentPart = SIZE_NONE;
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
      if (pending[cuDepth]) {
        team_wait(e, cuDepth); pending[cuDepth] = 0;
        const TeamBox *b = &HM_TEAM_PTR()->box[cuDepth];
        f->bestCost = b->cost; f->bestBits = b->bits; f->bestDist = b->dist;
      }
      if (f->splitCost < f->bestCost) {
        f->bestCost = f->splitCost; f->bestBits = f->splitBits; f->bestDist = f->splitDist;
        cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
      } else team_take_unsplit(e, cuDepth, cuZ, cuDepth);
      retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue;
    }
    if (f->phase == 2) { // a sub-CU returned
      f->splitBits += retBits; f->splitDist += retDist; f->phase = 1;
      // the exact early stop of compress_ctu (same conditions), applied when the unsplit result is already there
      if (pending[cuDepth] && team_ready(e, cuDepth)) {
        pending[cuDepth] = 0;
        const TeamBox *b = &HM_TEAM_PTR()->box[cuDepth];
        f->bestCost = b->cost; f->bestBits = b->bits; f->bestDist = b->dist;
      }
      bool laterSibling = sp == 0;
      if (sp > 0) {
        const int pq = parts, pz = fr[sp - 1].cuZ;
        for (int s2 = fr[sp - 1].sub; s2 < 4; s2++) {
          const int r2 = hm_z2r(pz + s2 * pq);
          laterSibling |= (e->ctuX * 64 + (r2 & 15) * 4 < e->width) && (e->ctuY * 64 + (r2 >> 4) * 4 < e->height);
        }
      }
      if (!pending[cuDepth] && f->sub < 4 && laterSibling && !f->boundary && !(calc_rd_cost(e, f->splitBits, f->splitDist) < f->bestCost)) {
        team_take_unsplit(e, cuDepth, cuZ, cuDepth);
        retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--;
      }
      continue;
    }
  }
  e->outCost = retCost; e->outBits = retBits; e->outDist = retDist;
}

// The mode tests of one CU in a P / B slice (compress_cu_inter_modes, hm355_inter_cu.h) by two wavefronts: this one takes the merge and 2Nx2N
// candidates, the horizontal AMP pair and the intra candidates, its partner (wave `mate`) Nx2N / 2NxN and the vertical AMP pair.  Every candidate
// starts from the same CABAC snapshot and the same neighbourhood; what they share is xCheckBestMode's running best, and that is put together
// here in the reference's order: this wavefront's candidates of a stage come first in it, so the partner's best replaces the running best only
// when it is strictly cheaper.  The partner's 2Nx2N-dependent input -- m_integerMv2Nx2N after this CU's 2Nx2N search -- comes from me_token_prepass.
// Returns the wave whose workspace holds the best mode (best[cuDepth], CI_NEXT_BEST of that depth).
HM_DEV HM_NOINLINE int compress_cu_inter_modes_duo(Shared *e, int cuZ, int cuDepth, int sp, int self, int mate, int mate2 = 0)
{ // mate2 != 0: a second partner for 2NxN (the first one then only takes Nx2N); CUs without AMP candidates only (the smallest CU size)
  HM_ENTRY(e); cuZ = HM_UNI(cuZ); cuDepth = HM_UNI(cuDepth); sp = HM_UNI(sp); self = HM_UNI(self); mate = HM_UNI(mate); mate2 = HM_UNI(mate2);
  if (HM_UNI(HM_TEAM_PTR()->waves) < HM_TEAM) { compress_cu_inter_modes(e, cuZ, cuDepth, sp); return self; }   // a launch of five-wavefront teams: the whole chain here
  CuFrame *f = &e->cuf[sp];
  WorkSpace *mw = team_wave_ws(mate);
  TeamBox *mb = &HM_TEAM_PTR()->box[mate - 1];
  int owner = self;
  f->ampSens = 0;
  { // the partner's Nx2N / 2NxN searches start from the integer vectors this CU's 2Nx2N search will leave; this wavefront's own 2Nx2N search from those it found
    const int i = hm_lane() & 31;
    const MvD keep = e->ws->intMv[i >> 4][i & 15];
    me_token_prepass(e, cuZ, cuDepth);
    HM_PAR_FOR(k, 32) mw->intMv[k >> 4][k & 15] = e->ws->intMv[k >> 4][k & 15];
    if (mate2) { WorkSpace *mw2 = team_wave_ws(mate2); HM_PAR_FOR(k, 32) mw2->intMv[k >> 4][k & 15] = e->ws->intMv[k >> 4][k & 15]; }
    HM_SYNC();
    if (hm_lane() < 32) e->ws->intMv[i >> 4][i & 15] = keep;
    HM_SYNC();
  }
  team_post(e, mate - 1, cuZ, cuDepth, mate2 ? 1 : 3, TK_PAIR1, self);
  if (mate2) team_post(e, mate2 - 1, cuZ, cuDepth, 2, TK_PAIR1, self);
  check_rd_cost_merge_2Nx2N(e, cuZ, cuDepth, sp);
  check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2Nx2N, 0, sp);
  team_wait(e, mate - 1);
  if (HM_UNI(mb->improved) && mb->cost < f->bestCost) { f->bestCost = mb->cost; f->bestBits = mb->bits; f->bestDist = mb->dist; owner = mate; }
  if (mate2) {                          // 2NxN follows Nx2N in the reference's order
    team_wait(e, mate2 - 1);
    mb = &HM_TEAM_PTR()->box[mate2 - 1];
    if (HM_UNI(mb->improved) && mb->cost < f->bestCost) { f->bestCost = mb->cost; f->bestBits = mb->bits; f->bestDist = mb->dist; owner = mate2; }
  }
  // TEncBinCABACCounter::resetBits keeps the fractional bits (TEncBinCoderCABAC.cpp:161), so what follows the candidates on the go-on coder "as it
  // stands" (the split flags, TEncCu.cpp:859-863, :1042-1047) sees the remainder the LAST candidate in the reference's order left -- the partner's, here
  int mateLast = 1; uint32_t mateFrac = (uint32_t)HM_UNI(mb->sens);
  mb = &HM_TEAM_PTR()->box[mate - 1];
  if (cuDepth < 3) { // deriveTestModeAMP :386-447 on the best mode so far
    const Best *b = &team_wave_ws(owner)->best[cuDepth];
    const int ps = HM_UNI(b->m.part[cuZ]), bmrg = HM_UNI(b->im.mrg[cuZ]), bskip = HM_UNI(b->im.skip[cuZ]), parent = f->parentPart;
    int hor = 0, ver = 0, mh = 0, mv = 0;
    if (ps == SIZE_2NxN) hor = 1;
    else if (ps == SIZE_Nx2N) ver = 1;
    else if (ps == SIZE_2Nx2N && !bmrg && !bskip) { hor = 1; ver = 1; }
    if (parent >= SIZE_2NxnU && parent <= SIZE_nRx2N) { mh = 1; mv = 1; }
    if (parent == SIZE_NONE) { if (ps == SIZE_2NxN) mh = 1; else if (ps == SIZE_Nx2N) mv = 1; }
    if (ps == SIZE_2Nx2N && !bskip) { mh = 1; mv = 1; }
    if ((64 >> cuDepth) == 64) { hor = 0; ver = 0; }
    f->ampSens = (int8_t)((!hor && !mh) || (!ver && !mv));
    const int mateBusy = ver || mv;
    if (mateBusy) team_post(e, mate - 1, cuZ, cuDepth, ver ? 1 : 2, TK_PAIR2, self, f->bestCost);
    const double before = f->bestCost;
    if (hor) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnU, 0, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnD, 0, sp); }
    else if (mh) { check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnU, 1, sp); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxnD, 1, sp); }
    if (f->bestCost < before) owner = self;
    if (hor || mh) mateLast = 0;
    if (mateBusy) {
      team_wait(e, mate - 1);
      if (HM_UNI(mb->improved) && mb->cost < f->bestCost) { f->bestCost = mb->cost; f->bestBits = mb->bits; f->bestDist = mb->dist; owner = mate; }
      mateLast = 1; mateFrac = (uint32_t)HM_UNI(mb->sens);
    }
  }
  { // intra only when the best inter mode left a residual (:820)
    const Best *b = &team_wave_ws(owner)->best[cuDepth];
    if (HM_UNI(b->m.cbf[0][cuZ]) != 0 || HM_UNI(b->m.cbf[1][cuZ]) != 0 || HM_UNI(b->m.cbf[2][cuZ]) != 0) {
      const double before = f->bestCost;
      check_rd_cost_intra(e, cuZ, cuDepth, SIZE_2Nx2N); check_best_mode(e, f, cuZ, cuDepth);
      if (cuDepth == 3) { check_rd_cost_intra(e, cuZ, cuDepth, SIZE_NxN); check_best_mode(e, f, cuZ, cuDepth); }
      if (f->bestCost < before) owner = self;
      mateLast = 0;
    }
  }
  if (mateLast) { if (hm_lane() == 0) e->cur.frac = (e->cur.frac & ~32767ull) | mateFrac; HM_SYNC(); }
  return owner;
}

// TEncCu::compressCtu -> xCompressCU for a P / B slice: the unsplit CUs of depth 0..2 with the helpers, the 8x8 CUs here (see the head of this file)
HM_DEV HM_NOINLINE void compress_ctu_team_inter(Shared *e)
{
  HM_ENTRY(e);
  HM_TSTAT_T0(tAll);
  CtuMeta *m = (&e->meta);
  CuFrame *fr = e->cuf; int sp = 0;
  fr[0].cuZ = 0; fr[0].phase = 0; fr[0].parentPart = SIZE_NONE;
  double retCost = 0; uint32_t retBits = 0, retDist = 0;
  int pending[3] = {0, 0, 0};          // the unsplit CU of this depth is with its helper
  int guessed[3] = {0, 0, 0};          // a sub-CU of this depth's CU started before the helper had answered: its parent part size was taken as "not AMP"
  int sensitive[3] = {0, 0, 0};        // ... and its candidate list would have been another one under an AMP parent
  while (sp >= 0) {
    CuFrame *f = &fr[sp]; const int cuDepth = sp, cuZ = f->cuZ;
    const int size = 64 >> cuDepth, parts = 256 >> (2 * cuDepth), q = parts >> 2;
    if (f->phase == 0) {
      const int r = hm_z2r(cuZ);
      const int lx = e->ctuX * 64 + (r & 15) * 4, ty = e->ctuY * 64 + (r >> 4) * 4;
      f->boundary = !((lx + size - 1 < e->width) && (ty + size - 1 < e->height));
      f->bestCost = HM_MAX_DOUBLE; f->bestBits = 0; f->bestDist = 0;
      if (cuDepth == 3) {               // TEncCu.cpp:628-863 at the smallest CU size, as compress_ctu does it
        if (!f->boundary) {
          HM_TSTAT_T0(t8);
          const int owner = compress_cu_inter_modes_duo(e, cuZ, cuDepth, sp, 0, 4, 8);
          HM_TSTAT_T1(3, t8);
          reset_bits(&e->cur);
          f->bestBits += num_bits(&e->cur);
          f->bestCost = calc_rd_cost(e, f->bestBits, f->bestDist);
          if (owner != 0) {              // the partner's candidate won: its workspace holds the CU (as team_take_unsplit)
            WorkSpace *ow = team_wave_ws(owner);
            restore_best_from(e, &ow->best[cuDepth], cuZ, cuDepth);
            cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &ow->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)]);
          } else restore_best(e, cuZ, cuDepth);
        } else restore_best(e, cuZ, cuDepth);
        retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue;
      }
      pending[cuDepth] = 0; guessed[cuDepth] = 0; sensitive[cuDepth] = 0;
      if (!f->boundary) {
        WorkSpace *hw = team_helper_ws(e, cuDepth);
        HM_PAR_FOR(i, 32) hw->intMv[i >> 4][i & 15] = e->ws->intMv[i >> 4][i & 15];   // m_integerMv2Nx2N as the unsplit CU finds it
        HM_SYNC();
        team_post(e, cuDepth, cuZ, cuDepth, f->parentPart);
        pending[cuDepth] = 1;
        me_token_prepass(e, cuZ, cuDepth);                                             // ... and as it leaves it
        HM_PAR_FOR(i, 32) e->ws->teamTok[cuDepth][i >> 4][i & 15] = e->ws->intMv[i >> 4][i & 15];
        HM_SYNC();
      }
      init_est_data(e, cuZ, cuDepth);
      f->splitBits = 0; f->splitDist = 0; f->sub = 0; f->phase = 1;
    }
    if (f->phase == 1) {
      if (f->sub < 4) {
        const int s = f->sub++;
        const int subZ = cuZ + s * q, r = hm_z2r(subZ);
        const int sx = e->ctuX * 64 + (r & 15) * 4, sy = e->ctuY * 64 + (r >> 4) * 4;
        HM_PAR_FOR(i, q) { m->depth[subZ + i] = (uint8_t)(cuDepth + 1); m->part[subZ + i] = SIZE_NONE; m->pred[subZ + i] = MODE_NONE; }
        HM_SYNC();
        if (sx < e->width && sy < e->height) {
          if (s == 0) cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
          else cabac_copy(&e->ws->slot[HM_SLOT(cuDepth + 1, CI_CURR_BEST)], &e->ws->slot[HM_SLOT(cuDepth + 1, CI_NEXT_BEST)]);
          fr[sp + 1].cuZ = (int16_t)subZ; fr[sp + 1].phase = 0;
          // AMP speed-up: the part size of this depth's best mode when it is inter (TEncCu.cpp:1026) -- known once the helper has answered
          int parentPart = SIZE_NONE;
          if (!f->boundary) {
            if (pending[cuDepth] && team_ready(e, cuDepth)) {
              pending[cuDepth] = 0;
              const TeamBox *b = &HM_TEAM_PTR()->box[cuDepth];
              f->bestCost = b->cost; f->bestBits = b->bits; f->bestDist = b->dist;
            }
            if (!pending[cuDepth]) {
              const Best *hb = &team_answer_ws(e, cuDepth)->best[cuDepth];
              if (HM_UNI(hb->m.pred[cuZ]) == MODE_INTER) parentPart = HM_UNI(hb->m.part[cuZ]);
            } else guessed[cuDepth] = 1;
          }
          fr[sp + 1].parentPart = (int8_t)parentPart;
          f->phase = 2; sp++; continue;
        }
        continue;
      }
      if (pending[cuDepth]) {
        HM_TSTAT_T0(tw);
        team_wait(e, cuDepth); pending[cuDepth] = 0;
        HM_TSTAT_T1(2, tw);
        const TeamBox *b = &HM_TEAM_PTR()->box[cuDepth];
        f->bestCost = b->cost; f->bestBits = b->bits; f->bestDist = b->dist;
      }
      if (!f->boundary && guessed[cuDepth] && sensitive[cuDepth]) {
        const Best *hb = &team_answer_ws(e, cuDepth)->best[cuDepth];
        const int pp = HM_UNI(hb->m.pred[cuZ]) == MODE_INTER ? HM_UNI(hb->m.part[cuZ]) : SIZE_NONE;
        if (pp >= SIZE_2NxnU && pp <= SIZE_nRx2N) {
          // the guess was wrong where it mattered: the sub-CUs again, from the state the unsplit CU's 2Nx2N search left (nothing outside this CU has read them)
          HM_PAR_FOR(i, 32) e->ws->intMv[i >> 4][i & 15] = e->ws->teamTok[cuDepth][i >> 4][i & 15];
          HM_SYNC();
          guessed[cuDepth] = 0; sensitive[cuDepth] = 0;
          HM_TSTAT_ADD(4, 1);
          init_est_data(e, cuZ, cuDepth);
          f->splitBits = 0; f->splitDist = 0; f->sub = 0;
          continue;
        }
      }
      if (!f->boundary) {
        reset_bits(&e->cur);
        enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), m->depth[cuZ] > cuDepth);
        f->splitBits += num_bits(&e->cur);
      }
      f->splitCost = calc_rd_cost(e, f->splitBits, f->splitDist);
      cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)], &e->ws->slot[HM_SLOT(cuDepth + 1, CI_NEXT_BEST)]);
      if (f->splitCost < f->bestCost) {
        f->bestCost = f->splitCost; f->bestBits = f->splitBits; f->bestDist = f->splitDist;
        cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
      } else team_take_unsplit(e, cuDepth, cuZ, cuDepth);
      if (sp > 0 && !f->boundary) sensitive[sp - 1] |= HM_UNI(HM_TEAM_PTR()->box[cuDepth].sens);
      retCost = f->bestCost; retBits = f->bestBits; retDist = f->bestDist; sp--; continue;
    }
    if (f->phase == 2) { f->splitBits += retBits; f->splitDist += retDist; f->phase = 1; continue; }   // a sub-CU returned
  }
  HM_TSTAT_T1(1, tAll); HM_TSTAT_ADD(0, 1);
  e->outCost = retCost; e->outBits = retBits; e->outDist = retDist;
}

// ---- helper wavefront `wave` (1 .. HM_TEAM - 1) ----
// the uniform per-CTU context of the requesting wavefront, with private places for everything a candidate writes
HM_DEV inline void team_adopt(Shared *e, const Shared *src, int wave, Pel *win)
{
  e->P = src->P; e->fb = src->fb; e->tab = src->tab;
  e->width = src->width; e->height = src->height; e->bitDepth = src->bitDepth; e->wCtu = src->wCtu;
  for (int c = 0; c < 3; c++) e->stride[c] = src->stride[c];
  e->ctuX = src->ctuX; e->ctuY = src->ctuY; e->ctuAddr = src->ctuAddr;
  e->ws = team_wave_ws(wave); e->cc = e->ws->teamCoef; e->im = (InterMeta *)0; e->mpmZ = -1; e->s8Reuse = 0;
  if (e->fb.dqp && hm_lane() == 0) e->ws->dq = src->ws->dq;    // QP of the CTU, its predictor (m_bEncodeDQP is clear whenever a team searches)
  if (win) {
    // the window: rows [ctuY * S - 1, ctuY * S + S) of a plane with the picture's stride, addressed like the picture
    Pel *w = win;
    for (int c = 0; c < 3; c++) {
      const int S = c ? 32 : 64;
      e->fb.rec[c] = w - ((long long)e->ctuY * S - 1) * e->stride[c];
      w += (size_t)(S + 1) * e->stride[c];
    }
  }
  HM_SYNC();
}
HM_DEV inline void team_fill_window(Shared *e, const Shared *mainSh)
{ // everything a CU of this CTU can read outside itself (TComPattern.cpp:107-165): the row above up to 2 CTU widths, the column to the left, the CTU
  for (int c = 0; c < 3; c++) {
    const int S = c ? 32 : 64, ps = e->stride[c];
    const int x0 = e->ctuX * S > 0 ? e->ctuX * S - 1 : 0, x1 = (e->ctuX + 2) * S < ps ? (e->ctuX + 2) * S : ps;
    const int y0 = e->ctuY > 0 ? e->ctuY * S - 1 : 0, y1 = (e->ctuY + 1) * S, w = x1 - x0;
    const Pel *src = mainSh->fb.rec[c]; Pel *dst = e->fb.rec[c];
    HM_PAR_FOR_XY(x, y, w, w * (y1 - y0)) dst[(size_t)(y0 + y) * ps + x0 + x] = src[(size_t)(y0 + y) * ps + x0 + x];
  }
  HM_SYNC();
}
// the decision (and motion) arrays as the requesting wavefront holds them: everything outside the CU is final (inside, init_est_data starts afresh)
HM_DEV inline void team_copy_arrays(Shared *e, const Shared *src)
{
  { const uint32_t *s_ = (const uint32_t *)&src->meta; uint32_t *d_ = (uint32_t *)&e->meta;
    HM_PAR_FOR(i, (int)(sizeof(CtuMeta) / 4)) d_[i] = s_[i]; }
  if (src->im) {
    e->im = &e->ws->teamIm;
    const uint32_t *s_ = (const uint32_t *)src->im; uint32_t *d_ = (uint32_t *)e->im;
    HM_PAR_FOR(i, (int)(sizeof(InterMeta) / 4)) d_[i] = s_[i];
  }
  HM_SYNC();
}
HM_DEV inline void team_answer(TeamBox *b, uint32_t seq, double cost, uint32_t bits, uint32_t dist, int sens, int owner, int improved)
{
  if (hm_lane() == 0) { b->cost = cost; b->bits = bits; b->dist = dist; b->sens = sens; b->owner = owner; b->improved = improved; }
  HM_TEAM_RELEASE();
  team_st(&b->doneSeq, seq);
}
HM_DEV inline void team_helper(Team *T, int wave, Pel *win)
{
  Shared *e = &T->sh[wave]; const Shared *mainSh = &T->sh[0];
  TeamBox *b = &T->box[wave - 1];
  uint32_t seen = 0;
  for (;;) {
    uint32_t s;
    for (;;) {
      s = team_ld(&b->reqSeq);
      if (s != seen) break;
      if (team_ld(&T->quit)) return;
      __builtin_amdgcn_s_sleep(2);
    }
    HM_TEAM_ACQUIRE();
    seen = s;
    const int cuZ = HM_UNI(b->cuZ), cuDepth = HM_UNI(b->depth), part = HM_UNI(b->part), kind = HM_UNI(b->kind);
    const Shared *src = &T->sh[HM_UNI(b->src)];
    CuFrame *f = &e->cuf[cuDepth];
    if (kind == TK_PAIR2) {
      // the vertical AMP pair of the CU this wavefront just searched Nx2N / 2NxN for (its context is in place); a candidate only counts below the best cost so far
      const double thr = b->threshold;
      f->bestCost = thr; f->bestBits = 0; f->bestDist = 0;
      const int mrgOnly = (part & 1) ? 0 : 1;
      check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nLx2N, mrgOnly, cuDepth); check_rd_cost_inter(e, cuZ, cuDepth, SIZE_nRx2N, mrgOnly, cuDepth);
      team_answer(b, s, f->bestCost, f->bestBits, f->bestDist, (int)(e->cur.frac & 32767), wave, f->bestCost < thr);
      continue;
    }
    team_adopt(e, src, wave, win);
    team_copy_arrays(e, src);
    cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)], &src->ws->slot[HM_SLOT(cuDepth, CI_CURR_BEST)]);
    f->cuZ = (int16_t)cuZ; f->boundary = 0; f->parentPart = SIZE_NONE; f->ampSens = 0;
    f->bestCost = HM_MAX_DOUBLE; f->bestBits = 0; f->bestDist = 0;
    if (kind == TK_PAIR1) {
      // Nx2N and 2NxN of the requester's CU (m_integerMv2Nx2N as its 2Nx2N search leaves it is in this wavefront's workspace)
      if (part & 1) check_rd_cost_inter(e, cuZ, cuDepth, SIZE_Nx2N, 0, cuDepth);
      if (part & 2) check_rd_cost_inter(e, cuZ, cuDepth, SIZE_2NxN, 0, cuDepth);
      team_answer(b, s, f->bestCost, f->bestBits, f->bestDist, (int)(e->cur.frac & 32767), wave, f->bestCost < HM_MAX_DOUBLE);
      continue;
    }
    team_fill_window(e, mainSh);
    if (kind == TK_CHAIN) {
      // P / B slice: the whole candidate chain of the unsplit CU (TEncCu.cpp:628-863) on a private copy of the CTU's motion arrays; everything
      // outside the CU is final, m_integerMv2Nx2N was put into this wavefront's workspace by the main one
      f->parentPart = (int8_t)part;
      const int owner = compress_cu_inter_modes_duo(e, cuZ, cuDepth, cuDepth, wave, wave + 4);
      reset_bits(&e->cur);
      if (cuDepth != 3) enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
      const uint32_t bits = f->bestBits + num_bits(&e->cur), dist = f->bestDist;
      team_answer(b, s, calc_rd_cost(e, bits, dist), bits, dist, f->ampSens, owner, 1);
      continue;
    }
    // I slice: xCheckRDCostIntra + xCheckBestMode against an empty best (TEncCu.cpp:706-735, :1574, :1702)
    check_rd_cost_intra(e, cuZ, cuDepth, part);
    double c = e->outCost; uint32_t bits = e->outBits; const uint32_t dist = e->outDist;
    save_best(e, cuZ, cuDepth);
    cabac_copy(&e->ws->slot[HM_SLOT(cuDepth, CI_NEXT_BEST)], &e->ws->slot[HM_SLOT(cuDepth, CI_TEMP_BEST)]);
    if (cuDepth != 3) { // split flag of the unsplit candidate, TEncCu.cpp:859-863 (coded on the go-on coder as the candidate left it)
      reset_bits(&e->cur);
      enc_bin(e, &e->cur, C_SPLIT + ctx_split_flag(e, cuZ, cuDepth), 0);
      bits += num_bits(&e->cur);
      c = calc_rd_cost(e, bits, dist);
    }
    team_answer(b, s, c, bits, dist, 0, wave, 1);
  }
}
