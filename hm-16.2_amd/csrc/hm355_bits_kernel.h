// hm355 bitstream pass, part 2: TEncSlice::encodeSlice (TEncSlice.cpp:910-1095) on what the search (and SAO) left in a picture slot.
// One wavefront codes one substream (the whole slice, or one CTU row under WPP): context initialisation or the WPP hand-off from the
// row above, per CTU the SAO syntax (TEncSbac::codeSAOBlkParam :1674) and encode_ctu on the arithmetic coder (hm355_bits.h), then the
// terminating bin, TEncBinCABAC::finish and the byte alignment.  The wavefront of the last substream also evaluates
// TEncSbac::determineCabacInitIdx (:163-222).  Substreams of a picture depend on each other only through the WPP context hand-off
// (after the second CTU of the row above), published through a flag per row.
#pragma once

struct BitsParams {                   // one picture of the batch
  int32_t sliceType, qp, cabacInitType, saoEnabled[3];
  const int32_t *sao;                 // coded SAO parameters [numCtus][3][35] (mode, type, aux, offset[32]); NULL = no SAO syntax
  uint8_t *raw; uint32_t capPerCtu;   // substream k is written at raw + firstCtu(k) * capPerCtu
  uint8_t *packed;                    // the substreams back to back (hm355_bits_pack_kernel)
  uint32_t *subSizes;                 // [numSubstreams] bytes
  CabacW *sync;                       // [hCtu] contexts after the 2nd CTU of each CTU row (m_entropyCodingSyncContextState)
  uint32_t *syncFlag;                 // [hCtu] == epoch once sync[row] is published
  uint32_t *sched;                    // per launch, shared by the pictures of the batch: [0] ticket counter, [1] abort word
  uint32_t epoch;
  int32_t nextInitType; uint32_t bins; int32_t overflow;    // results
};

HM_CONST uint8_t HM_SAO_INIT[3][2] = { {153, 160}, {153, 185}, {153, 200} };   // [B, P, I][merge, type]: INIT_SAO_MERGE_FLAG / INIT_SAO_TYPE_IDX, ContextTables.h
HM_CONST double HM_STATE_TO_PROB_LPS[64] = {    // ContextModel3DBuffer::calcCost, ContextModel3DBuffer.cpp:97
  0.50000000, 0.47460857, 0.45050660, 0.42762859, 0.40591239, 0.38529900, 0.36573242, 0.34715948, 0.32952974, 0.31279528, 0.29691064, 0.28183267,
  0.26752040, 0.25393496, 0.24103941, 0.22879875, 0.21717969, 0.20615069, 0.19568177, 0.18574449, 0.17631186, 0.16735824, 0.15885931, 0.15079198,
  0.14313433, 0.13586556, 0.12896592, 0.12241667, 0.11620000, 0.11029903, 0.10469773, 0.09938088, 0.09433404, 0.08954349, 0.08499621, 0.08067986,
  0.07658271, 0.07269362, 0.06900203, 0.06549791, 0.06217174, 0.05901448, 0.05601756, 0.05317283, 0.05047256, 0.04790942, 0.04547644, 0.04316702,
  0.04097487, 0.03889405, 0.03691890, 0.03504406, 0.03326442, 0.03157516, 0.02997168, 0.02844963, 0.02700488, 0.02563349, 0.02433175, 0.02309612,
  0.02192323, 0.02080991, 0.01975312, 0.01875000 };

HM_DEV inline int bits_ctx_init_value(int i, int initType)
{ return i < HM_NUM_CTX ? (initType == 2 ? HM_CTX_INIT_I[i] : (initType == 1 ? HM_CTX_INIT_P[i] : HM_CTX_INIT_B[i])) : HM_SAO_INIT[initType][i - HM_NUM_CTX]; }
HM_DEV inline int bits_ctx_init_state(int iv, int qp)
{ // ContextModel::init, ContextModel.cpp:55-64
  qp = hm_clip3(0, 51, qp);
  const int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
  const int mps = st >= 64;
  return ((mps ? (st - 64) : (63 - st)) << 1) + mps;
}

// ---- SAO syntax (TEncSbac.cpp:1530-1709) ----
HM_DEV inline void bits_sao_max_uvlc(CabacW *w, int code, int maxSymbol)
{ // codeSaoMaxUvlc :1535: truncated unary in bypass bins
  if (maxSymbol == 0) return;
  if (code == 0) { enc_epv(w, 0, 1); return; }
  const int n = code + (maxSymbol > code ? 1 : 0);                 // `code` ones, then a zero unless the maximum is reached
  enc_epv(w, ((1u << code) - 1) << (maxSymbol > code ? 1 : 0), n);
}
HM_DEV inline void bits_sao_offset_param(const Shared *e, CabacW *w, int comp, const int32_t *p, int sliceEnabled, int maxOffQ)
{ // codeSAOOffsetParam :1597-1671; mode 0 = off, 1 = new, 2 = merge; type 0..3 = edge classes, 4 = band
  if (!sliceEnabled) return;
  const int first = comp != 2, mode = p[0], type = p[1], aux = p[2];
  if (first) {
    const int sym = mode == 0 ? 0 : (type == 4 ? 1 : 2);
    if (sym == 0) enc_bin(e, w, C_SAO_TYPE, 0); else { enc_bin(e, w, C_SAO_TYPE, 1); enc_epv(w, sym == 1 ? 0 : 1, 1); }
  }
  if (mode == 1) {
    int offset[4], k = 0;
    const int numClasses = type == 4 ? 4 : 5;
    for (int i = 0; i < numClasses; i++) {
      if (type != 4 && i == 2) continue;
      offset[k++] = p[3 + (type == 4 ? ((aux + i) & 31) : i)];
    }
    for (int i = 0; i < 4; i++) bits_sao_max_uvlc(w, hm_abs(offset[i]), maxOffQ);
    if (type == 4) { for (int i = 0; i < 4; i++) if (offset[i] != 0) enc_epv(w, offset[i] < 0, 1); enc_epv(w, (uint32_t)aux, 5); }
    else if (first) enc_epv(w, (uint32_t)type, 2);
  }
}
HM_DEV inline void bits_sao_blk_param(const Shared *e, CabacW *w, const int32_t *blk, const int32_t *sliceEnabled, int leftAvail, int aboveAvail, int maxOffQ)
{ // codeSAOBlkParam :1674-1709; merge = mode 2 with type 0 (left) / 1 (above)
  int isLeft = 0, isAbove = 0;
  if (leftAvail) { isLeft = blk[0] == 2 && blk[1] == 0; enc_bin(e, w, C_SAO_MERGE, isLeft); }
  if (aboveAvail && !isLeft) { isAbove = blk[0] == 2 && blk[1] == 1; enc_bin(e, w, C_SAO_MERGE, isAbove); }
  if (!isLeft && !isAbove) for (int c = 0; c < 3; c++) bits_sao_offset_param(e, w, c, blk + c * 35, sliceEnabled[c], maxOffQ);
}

// one substream of one picture.  Returns non-zero when a WPP wait was abandoned (device only).
HM_DEV inline int bits_encode_substream(Shared *e, CabacW *w, const Params *P, int frame, BitsParams *bp, int sub, int wsIndex)
{
  // uniform context (every lane writes the same values), as process_ctu sets it up for the search
  e->P = P; e->fb = P->frames[frame]; e->ws = P->ws + wsIndex; e->tab = P->tab;
  e->width = P->width; e->height = P->height; e->bitDepth = P->bitDepth; e->wCtu = P->wCtu; e->mpmZ = -1;
  for (int c = 0; c < 3; c++) e->stride[c] = P->stride[c];
  HM_SYNC();
  const int wpp = P->wpp, wCtu = P->wCtu, numCtus = wCtu * P->hCtu;
  const int first = wpp ? sub * wCtu : 0, last = wpp ? first + wCtu : numCtus;
  const int initType = bp->sliceType == 2 ? 2 : bp->cabacInitType;     // TEncSbac::resetEntropy :106-115
  const int qp = bp->qp;
  const int maxOffQ = (1 << ((P->bitDepth < 10 ? P->bitDepth : 10) - 5)) - 1;   // g_saoMaxOffsetQVal
  const int saoOn = bp->sao && (bp->saoEnabled[0] || bp->saoEnabled[1]);
  // start of the substream: resetEntropy (context init + TEncBinCABAC::start); a WPP row then takes the contexts stored after the
  // second CTU of the row above (TEncSlice.cpp:975-994)
  HM_PAR_FOR(i, 192) { w->s[i] = (uint8_t)(i < HM_NUM_CTX + 2 ? bits_ctx_init_state(bits_ctx_init_value(i, initType), qp) : 0); w->used[i] = 0; }
  cabw_load_tables(w);
  cabw_start(w);
  w->held = 0; w->numHeld = 0; w->len = 0; w->bins = 0;
  w->out = bp->raw + (size_t)first * bp->capPerCtu; w->cap = (uint32_t)(last - first) * bp->capPerCtu;
  HM_SYNC();
  if (wpp && sub > 0 && wCtu > 1) {
#ifndef HM355_HOSTSIM
    int bad = 0;
    if (hm_lane() == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(bp->syncFlag + (sub - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != bp->epoch) {
        __builtin_amdgcn_s_sleep(16);
        if (__hip_atomic_load(bp->sched + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() - t0 > 10ull * 100000000ull) {
          __hip_atomic_store(bp->sched + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); bad = 1; break;   // every workgroup of the launch drains
        }
      }
    }
    if (__shfl(bad, 0, 64)) return 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    const CabacW *src = bp->sync + (sub - 1);
    HM_PAR_FOR(i, 192) { w->s[i] = src->s[i]; w->used[i] = src->used[i]; }
    HM_SYNC();
  }
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  for (int i = 0; i < HM_PROF_N; i++) { e->prof[i] = 0; e->profCnt[i] = 0; }
#endif
  for (int a = first; a < last; a++) {
    HM_PROF_BEGIN(e, PR_TOTAL);
    e->ctuX = a % wCtu; e->ctuY = a / wCtu; e->ctuAddr = a;
    e->cc = e->fb.coef + (size_t)a * HM_COEF_CTU;
    e->im = e->fb.imeta ? e->fb.imeta + a : (InterMeta *)0;
    { // the decision arrays of the CTU into LDS, where the syntax functions read them
      const uint32_t *src = (const uint32_t *)(e->fb.meta + a); uint32_t *dst = (uint32_t *)&e->meta;
      HM_PAR_FOR(i, (int)(sizeof(CtuMeta) / 4)) dst[i] = src[i];
      HM_SYNC();
    }
    if (saoOn) bits_sao_blk_param(e, w, bp->sao + (size_t)a * 105, bp->saoEnabled, e->ctuX > 0, e->ctuY > 0, maxOffQ);
    if (e->fb.dqp) { // cu_qp_delta: the QP the search gave this CTU and its predictor; TEncCu::encodeCtu :358-361 sets m_bEncodeDQP
      const CtuDqp o = e->fb.dqp->out[a];
      if (hm_lane() == 0) { e->ws->dq.ctuQp = o.qp; e->ws->dq.refQp = o.refQp; e->ws->dq.flag = 1; }
      HM_SYNC();
    }
    { HM_PROF_BEGIN(e, PR_ENCCU); encode_ctu(e, w, a == numCtus - 1); HM_PROF_END(e, PR_ENCCU); }
    HM_PROF_END(e, PR_TOTAL);
    if (wpp && a == first + 1) { // m_entropyCodingSyncContextState.loadContexts, TEncSlice.cpp:1050
      HM_SYNC();
      CabacW *dst = bp->sync + sub;
      HM_PAR_FOR(i, 192) { dst->s[i] = w->s[i]; dst->used[i] = w->used[i]; }
#ifndef HM355_HOSTSIM
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (hm_lane() == 0) __hip_atomic_store(bp->syncFlag + sub, bp->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      HM_SYNC();
    }
  }
#if defined(HM355_PROFILE) && !defined(HM355_HOSTSIM)
  if (hm_lane() == 0 && P->prof) for (int i = 0; i < HM_PROF_N; i++) { atomicAdd(P->prof + i, e->prof[i]); atomicAdd(P->prof + HM_PROF_N + i, e->profCnt[i]); }
#endif
  // end of the substream (TEncSlice.cpp:1056-1072): end_of_subset_one_bit / end_of_slice_segment_flag = 1, flush, byte alignment
  enc_trm(e, w, 1);
  cabw_finish(w);
  cabw_put_bits(w, 1, 1);
  if (w->numHeld) cabw_put_bits(w, 0, 8 - w->numHeld);
  HM_SYNC();
  if (hm_lane() == 0) {
    bp->subSizes[sub] = w->len;
#ifndef HM355_HOSTSIM
    atomicAdd(&bp->bins, w->bins);
#else
    bp->bins += w->bins;
#endif
    if (w->len > w->cap) bp->overflow = 1;
  }
  if (last == numCtus) {
    // determineCabacInitIdx :163-222 on the contexts the slice ended with: for the B and the P table, the cost of every context that
    // coded a bin (ContextModel3DBuffer::calcCost); the cheaper table serves the following pictures, B on a tie
    int next = 2;
    if (bp->sliceType != 2) {
      uint32_t cost[2] = {0, 0};
      HM_PAR_FOR(i, HM_NUM_CTX + 2) {
        if (w->used[i]) {
          const int st = w->s[i];
          const double probLPS = HM_STATE_TO_PROB_LPS[st >> 1];
          double prob0, prob1;
          if (st & 1) { prob0 = probLPS; prob1 = 1.0 - prob0; } else { prob1 = probLPS; prob0 = 1.0 - prob1; }
          for (int t = 0; t < 2; t++) {
            const int is = bits_ctx_init_state(bits_ctx_init_value(i, t), qp);
            cost[t] += (uint32_t)(prob0 * (double)HM_ENTROPY_BITS[is ^ 0] + prob1 * (double)HM_ENTROPY_BITS[is ^ 1]);
          }
        }
      }
      cost[0] = hm_wave_sum(cost[0]); cost[1] = hm_wave_sum(cost[1]);
      next = cost[1] < cost[0] ? 1 : 0;
    }
    if (hm_lane() == 0) bp->nextInitType = next;
  }
  return 0;
}

#ifndef HM355_HOSTSIM
__shared__ CabacW g_cabw;
// Persistent grid over the (row, picture) items, handed out by ticket in row-major order (item = row * n + picture), exactly like the search
// kernel's scheduler: row r of a picture waits for the contexts row r-1 of the same picture publishes, which is item - n, a ticket that was
// taken earlier by a workgroup that is running (a workgroup only takes a ticket while it runs and never waits before taking one).  The oldest
// unfinished ticket therefore never waits, whatever the grid size and whatever else occupies the device: no residency assumption.
// The launch owns sched[0] (ticket) and sched[1] (abort), both zeroed by the host in the stream before every launch.
extern "C" __global__ void __launch_bounds__(64) hm355_bits_kernel(const Params *P, BitsParams *bps, int n, unsigned int *sched)
{
  const int numSub = P->wpp ? P->hCtu : 1, total = numSub * n;
  for (;;) {
    // the ticket travels from lane 0 to the wavefront through v_readfirstlane: a scalar value, so the scheduler loop and everything that
    // depends on (row, picture) stay wave-uniform for the compiler as well (handing it over through LDS made the loop "divergent": see DESIGN.md)
    int item = 0;
    if (threadIdx.x == 0) item = (int)atomicAdd(&sched[0], 1u);
    item = __builtin_amdgcn_readfirstlane(item);
    if (item >= total) break;
    const int sub = item / n, f = item - sub * n;
    if (__builtin_amdgcn_readfirstlane(bits_encode_substream(&g_sh, &g_cabw, P, f, bps + f, sub, (int)blockIdx.x))) break;
    __syncthreads();
  }
}
// substreams of a picture back to back: one workgroup per (substream, picture); nothing is packed after an abandoned launch
extern "C" __global__ void __launch_bounds__(64) hm355_bits_pack_kernel(const Params *P, BitsParams *bps, const unsigned int *sched)
{
  if (sched[1] != 0u) return;
  const int sub = (int)blockIdx.x, wCtu = P->wCtu;
  const BitsParams *bp = bps + blockIdx.y;
  const uint32_t cap = (uint32_t)(P->wpp ? wCtu : wCtu * P->hCtu) * bp->capPerCtu;
  uint32_t off = 0;
  for (int k = 0; k < sub; k++) off += bp->subSizes[k] < cap ? bp->subSizes[k] : cap;
  uint32_t len = bp->subSizes[sub]; if (len > cap) len = cap;       // an overflowing substream is reported through bp->overflow
  const uint8_t *src = bp->raw + (size_t)(P->wpp ? sub * wCtu : 0) * bp->capPerCtu;
  for (uint32_t i = threadIdx.x; i < len; i += 64) bp->packed[off + i] = src[i];
}
#endif
